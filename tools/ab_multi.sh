cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2; do
  echo "base $(run A=1)  W1_LDS=48 $(run HULC_W1_LDS=48)  W1_LDS=78 $(run HULC_W1_LDS=78)  C1_LDS=52 $(run HULC_C1_LDS=52)  UNPACK_Y=8 $(run HULC_UNPACK_Y=8)  UNPACK_Y=4 $(run HULC_UNPACK_Y=4)"
done
