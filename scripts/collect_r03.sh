# copy the round-3 summaries from gpurun_out/final3 into profiles/
cd "$(dirname "$0")/.."
for f in gpurun_out/final3/r03_*; do cp "$f" profiles/; done
ls profiles | grep r03
