# pass size x lanes at 512^2 / 1024^2 (quick figures, one box)
cd $GRAFT_REPO_ROOT
run() { # size sersic walkers streams chunk
  echo -n "size $1 streams $4 chunk $5: "; CONFIGS="$1:$2:$3" timeout -k 10 200 bash tools/quick_bench.sh sweep --no-extras --opt streams=$4 --chunk $5 | head -1 | awk '{for(i=1;i<=NF;i++) if($i=="evals/s") print $(i-1)}'
}
if [ "$1" = "whole_rounds" ]; then
  run 1024 4 256 2 6; run 1024 4 256 3 4; run 1024 4 256 4 4; run 1024 4 256 2 4; run 1024 4 256 3 5
  run 512 2 1024 2 24; run 512 2 1024 3 16; run 512 2 1024 4 16; run 512 2 1024 2 16; run 512 2 1024 4 12
  exit 0
fi
run 1024 4 256 2 6; run 1024 4 256 2 12; run 1024 4 256 2 24; run 1024 4 256 1 6; run 1024 4 256 1 12; run 1024 4 256 1 24
run 512 2 1024 2 24; run 512 2 1024 2 64; run 512 2 1024 2 128; run 512 2 1024 1 24; run 512 2 1024 1 48; run 512 2 1024 1 96
