# per-side A/B of three libraries (tools/side_costs.py): new (grouped rasteriser), pow (serial, power tables), old
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/ab_parity.log 2>&1 || { tail -30 gpurun_out/ab_parity.log; exit 1; }
tail -2 gpurun_out/ab_parity.log
timeout -k 10 420 python3 tools/side_costs.py > gpurun_out/side_costs_new.jsonl
echo new done
PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_pow.so timeout -k 10 420 python3 tools/side_costs.py > gpurun_out/side_costs_pow.jsonl
echo pow done
PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_old.so timeout -k 10 420 python3 tools/side_costs.py > gpurun_out/side_costs_old.jsonl
echo old done
