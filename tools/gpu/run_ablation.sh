# ablation of the planner / fusion switches on the default workload (each line: one bench run, 10 steps after 3 warm-up)
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%-28s %7.1f clips/s  %6.2f ms/step' % ('$*', j['value'], j['ms_per_step']))"; }
run DIN_NONE=1
run DIN_CONV_FASTK=0
run DIN_CONV_KORDER=0
run DIN_CONV_HALO=0
run DIN_CONV_SMALL=0
run DIN_WGRAD_RING=0
run DIN_FUSE_1X1=0
run DIN_FUSE_FWD=0
run DIN_FUSE_WGRAD=0
run DIN_POOL_COMMUTE=0
run DIN_CONV_PIPE=4
run DIN_CONV_EPI_BATCH=0
run DIN_NONE=2
