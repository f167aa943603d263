# same-box A/B: the in-tree library against psfmc_amd/libpsfmc_old.so, alternating
cd $GRAFT_REPO_ROOT
export CONFIGS="${CONFIGS:-256:1:4096}"
for i in 1 2 3; do
  echo "== new"; timeout -k 10 300 bash tools/quick_bench.sh abh_new --no-extras | grep -E "evals/s|rows_fwd"
  echo "== old"; PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_old.so timeout -k 10 300 bash tools/quick_bench.sh abh_old --no-extras | grep -E "evals/s|rows_fwd"
done
