// Diagnostic build (never shipped): where a stage of conv3d_fwd_k3_persist spends its cycles.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCTU_STAMP -Iinclude -Ict-unet_amd/csrc scripts/diag_stamp.hip \
//         ct-unet_amd/csrc/elementwise.hip ct-unet_amd/csrc/convt.hip -o scripts/build/diag_stamp && gpurun_out/diag_stamp <cin_p> <nout_p> <size> <layout>
// Phases (wave 0 of every block, s_memtime cycles): 0 wait at the top barrier, 1 LDS write + barrier,
// 2 prefetch issue, 3 MFMA loop, 4 epilogue.
#include "conv3d.hip"
#include <vector>
#include <cstdlib>
#include <algorithm>

int main(int argc, char** argv) {
    const int cin_p = atoi(argv[1]), nout_p = atoi(argv[2]), S = atoi(argv[3]), layout = atoi(argv[4]);
    const size_t vox = (size_t)S * S * S;
    float *in, *out, *wp, *stats;
    hipMalloc(&in, vox * cin_p * 4); hipMalloc(&out, vox * nout_p * 4);
    const size_t nwp = ctu_conv3d_packed_floats(3, cin_p, nout_p, layout);
    hipMalloc(&wp, nwp * 4);
    std::vector<float> h(vox * cin_p);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(nwp);
    for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(wp, hw.data(), nwp * 4, hipMemcpyHostToDevice);
    const int nb = ctu_conv3d_num_blocks(1, S, S, S, 3, nout_p, layout);
    hipMalloc(&stats, (size_t)nb * 2 * nout_p * 4);
    unsigned long long* dbg;
    const size_t ndbg = (size_t)4096 * 6;
    hipMalloc(&dbg, ndbg * 8); hipMemset(dbg, 0, ndbg * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), &dbg, sizeof(dbg));
    if (argc > 5) {             // 6th argument: time conv3d_wgrad (in = activations, "out" buffer = upstream gradient)
        hipMemcpy(out, h.data(), std::min(h.size(), vox * nout_p) * 4, hipMemcpyHostToDevice);
        float *ws, *dw;
        hipMalloc(&ws, ctu_conv3d_wgrad_ws_floats(1, S, S, S, 3, cin_p, nout_p) * 4);
        hipMalloc(&dw, (size_t)cin_p * nout_p * 27 * 4);
        for (int it = 0; it < 5; ++it)
            if (ctu_conv3d_wgrad(in, cin_p, cin_p, nullptr, nullptr, 0, out, nout_p, nout_p, dw, nullptr, nout_p, cin_p, nullptr, ws,
                                 1, S, S, S, 3, nullptr)) { printf("error: %s\n", ctu_last_error()); return 1; }
    } else
    for (int it = 0; it < 5; ++it)
        if (ctu_conv3d_fwd(in, cin_p, cin_p, nullptr, nullptr, 0, wp, nullptr, 0, out, nout_p, nout_p, stats, 1, S, S, S, 3, layout, nullptr, nullptr)) {
            printf("error: %s\n", ctu_last_error()); return 1;
        }
    hipDeviceSynchronize();
    std::vector<unsigned long long> r(ndbg);
    hipMemcpy(r.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
    double ph[5] = {0, 0, 0, 0, 0}, ns = 0; int blocks = 0;
    for (size_t b = 0; b < 4096; ++b) if (r[b * 6 + 5]) { ++blocks; ns += r[b * 6 + 5]; for (int k = 0; k < 5; ++k) ph[k] += r[b * 6 + k]; }
    printf("cin_p %d nout_p %d %d^3 layout %d: %d blocks, %.1f stages/block\n", cin_p, nout_p, S, layout, blocks, ns / blocks);
    const char* nm[5] = {"wait@top-barrier", "LDS write+barrier", "prefetch issue", "MFMA loop", "epilogue"};
    double tot = 0; for (int k = 0; k < 5; ++k) tot += ph[k];
    for (int k = 0; k < 5; ++k) printf("  %-18s %9.0f cycles/stage  %5.1f %%\n", nm[k], ph[k] / ns, 100 * ph[k] / tot);
    printf("  total %.0f cycles/stage\n", tot / ns);
    return 0;
}
