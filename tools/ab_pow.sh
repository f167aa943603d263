# A/B of the rasteriser forms (round 3): prefetching kernels with grouped (g4) / serial (g1) pixels, serial without
# prefetch (pow), round-3 start (old); then the timing experiments on k_rows_fwd (dbg1: no stores, dbg2: no transform, dbg3: neither)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -2 gpurun_out/ab_parity.log
rm -f gpurun_out/pw5_*_quick.txt
export CONFIGS="512:2:1024 1024:4:256 256:2:4096 300:2:2048"
for v in g4 g1 pow old; do
echo "== $v"; PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_$v.so timeout -k 10 300 bash tools/quick_bench.sh pw5_$v --no-extras
done
export CONFIGS="1024:4:256 1024:0:256 512:2:1024 512:0:1024"
for v in g1 dbg1 dbg2 dbg3; do
echo "== $v"; PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_$v.so timeout -k 10 300 bash tools/quick_bench.sh pw5_$v --no-extras
done
