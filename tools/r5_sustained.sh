# Round 5, final code: long sustained legs of the sample pipeline (periodic whole-lattice comparisons on both lattices included; every
# sweep audited) at the three lattice sizes -> profiles/r05_sustained.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
O=gpurun_out/r5/sustained.txt; : > $O
for cfg in "256 both 1024" "128 both 2048" "64 hand 4096"; do
  set -- $cfg
  python bench.py --grid $1 --branches $2 --steps 8 --warmup 2 --sustained $3 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --details gpurun_out/r5/sustained_$1.json 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); c = b['config']; s = c['sweeps']
print('N=$1 $2: %d samples sustained %.3f ms/sample = %.2f meshes/s; whole-lattice comparisons in the leg %d; refused sweeps %d; in the run: %d audited sweeps, %d refused, tail ratios %.2f / %.2f, min tau / estimate %.2f' % (
    c['sustained_steps'], c['sustained_ms_per_step_incl_recalibration'], c['sustained_meshes_per_s'], c['sustained_recalibrations'], c['sustained_refused_sweeps'],
    s['audited'], s['refused'], s['tail_ratio_max'], s.get('zoom_lattice_tail_ratio_max') or 0.0, s['min_tau_over_estimate']))" >> $O
done
cat $O
