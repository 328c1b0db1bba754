// Diagnostic build (never shipped): where a box of lp_conv_fwd_p1_kernel (16-bit persistent forward) spends its cycles.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCTU_LP_STAMP -Iinclude -Ict-unet_amd/csrc scripts/diag_stamp_lp.hip \
//         ct-unet_amd/csrc/elementwise.hip -o gpurun_out/diag_stamp_lp && gpurun_out/diag_stamp_lp <cin_p> <nout_p> <size> [xf]
// Phases (wave 0 of every block, s_memtime cycles): 0 wait at the top barrier, 1 LDS write (+ transform) + barrier,
// 2 prefetch issue, 3 MFMA loop, 4 epilogue.
#include "conv3d_lp.hip"
#include <vector>
#include <cstdlib>

int main(int argc, char** argv) {
    const int cin_p = atoi(argv[1]), nout_p = atoi(argv[2]), S = atoi(argv[3]);
    const bool xf = argc > 4;
    const size_t vox = (size_t)S * S * S;
    void *in, *out, *wp;
    float *stats, *sc;
    hipMalloc(&in, vox * cin_p * 2); hipMalloc(&out, vox * nout_p * 2);
    hipMemset(in, 0x3c, vox * cin_p * 2);
    const size_t nwp = ctu_lp_conv3d_packed_elems(3, cin_p, nout_p);
    hipMalloc(&wp, nwp * 2); hipMemset(wp, 0x3b, nwp * 2);
    hipMalloc(&sc, 2 * 256 * 4);
    std::vector<float> h(512, 1.0f);
    hipMemcpy(sc, h.data(), 512 * 4, hipMemcpyHostToDevice);
    const int nb = ctu_lp_conv3d_num_blocks(1, S, S, S, 3, cin_p);
    hipMalloc(&stats, (size_t)nb * 2 * nout_p * 4);
    unsigned long long* dbg;
    const size_t ndbg = (size_t)8192 * 6;
    hipMalloc(&dbg, ndbg * 8); hipMemset(dbg, 0, ndbg * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_lp_stamp_out), &dbg, sizeof(dbg));
    for (int it = 0; it < 5; ++it)
        if (ctu_lp_conv3d_fwd(CTU_BF16, in, cin_p, cin_p, xf ? sc : nullptr, xf ? sc + 256 : nullptr, 1, wp, nullptr, 0, out, nout_p, nout_p,
                              stats, 1, S, S, S, 3, nullptr)) { printf("error: %s\n", ctu_last_error()); return 1; }
    hipDeviceSynchronize();
    std::vector<unsigned long long> r(ndbg);
    hipMemcpy(r.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost);
    double ph[5] = {0, 0, 0, 0, 0}, ns = 0; int blocks = 0;
    for (size_t b = 0; b < 8192; ++b) if (r[b * 6 + 5]) { ++blocks; ns += r[b * 6 + 5]; for (int k = 0; k < 5; ++k) ph[k] += r[b * 6 + k]; }
    printf("cin_p %d nout_p %d %d^3 xf %d: %d blocks, %.1f boxes/block\n", cin_p, nout_p, S, (int)xf, blocks, ns / blocks);
    const char* nm[5] = {"wait@top-barrier", "LDS write+barrier", "prefetch issue", "MFMA loop", "epilogue"};
    double tot = 0; for (int k = 0; k < 5; ++k) tot += ph[k];
    for (int k = 0; k < 5; ++k) printf("  %-18s %9.0f cycles/box  %5.1f %%\n", nm[k], ph[k] / ns, 100 * ph[k] / tot);
    printf("  total %.0f cycles/box\n", tot / ns);
    return 0;
}
