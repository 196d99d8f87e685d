cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--grid', '64', '--branches', 'hand', '--steps', '256', '--warmup', '8', '--no-cpu-baseline', '--no-other-math', '--no-other-sweeps', '--no-other-configs', '--sustained', '0']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
pr.dump_stats('gpurun_out/r5/prof64.pstats')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
open('gpurun_out/r5/prof64.txt', 'w').write(s.getvalue())
" > gpurun_out/r5/prof64_bench.json 2> gpurun_out/r5/prof64.err
tail -c 600 gpurun_out/r5/prof64_bench.json
head -70 gpurun_out/r5/prof64.txt
