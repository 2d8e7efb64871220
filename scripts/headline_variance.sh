cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do timeout 600 python bench.py --mode query --no-cpu-baseline > gpurun_out/var_$i.json 2> gpurun_out/var.err; done
for i in 1 2 3; do CHORE_BENCH_NOCAL=1 timeout 600 python bench.py --mode query --no-cpu-baseline > gpurun_out/var_nocal_$i.json 2> gpurun_out/var.err; done
