# same-box A/B of library builds: tools/ab_libs.sh <lib a> <lib b> (names under psfmc_amd/, without lib...so), alternating
cd $GRAFT_REPO_ROOT
A=$1; B=$2
export CONFIGS="${CONFIGS:-512:2:1024 1024:4:256}"
for i in 1 2 3; do
  for v in $A $B; do
    echo "== $v"; PSFMC_LIB=$GRAFT_REPO_ROOT/psfmc_amd/libpsfmc_$v.so timeout -k 10 300 bash tools/quick_bench.sh ab_$v --no-extras | grep -E "evals/s|rows_fwd"
  done
done
