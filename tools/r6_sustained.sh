# Round 6, final code: long legs of the sample pipeline under the DEFAULT (ordinary sweeps, every voxel) and under --fast (periodic
# whole-lattice comparisons on both lattices included; every sweep audited) at the three lattice sizes, and a soak of the short-list
# kernel's cluster form with its (recoverable) wait bound in place -> profiles/r06_sustained.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
O=gpurun_out/r6/sustained.txt; : > $O
ARGS="--no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs"
for cfg in "256 both 256" "128 both 1024" "64 hand 4096"; do
  set -- $cfg
  python bench.py --grid $1 --branches $2 --steps $3 --warmup 4 --sustained 0 $ARGS --details gpurun_out/r6/sustained_default_$1.json 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); c = b['config']; r = b['roofline']
d = json.load(open('gpurun_out/r6/sustained_default_$1.json'))['sweeps']
print('DEFAULT N=$1 $2: %d samples %.3f ms/sample = %.2f meshes/s; K1h %.3f ms per launch (%d launches timed), frac %.3f; ordinary sweeps coarse / fine %d / %d, refused-and-repeated %d, samples enqueued in one go %d' % (
    b['steps'], b['ms_per_step'], b['value'], r['launch_ms'], r['launches_timed'], r['frac'], d['coarse_sweeps']['exact'], d['fine_sweeps']['exact'], d['refused_sweeps'], d['samples_enqueued_in_one_go']))" >> $O
done
for cfg in "256 both 1024" "128 both 2048" "64 hand 4096"; do
  set -- $cfg
  python bench.py --fast --grid $1 --branches $2 --steps 8 --warmup 2 --sustained $3 $ARGS --details gpurun_out/r6/sustained_fast_$1.json 2>/dev/null | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); c = b['config']; s = c['sweeps']
print('--fast  N=$1 $2: %d samples sustained %.3f ms/sample = %.2f meshes/s; whole-lattice comparisons in the leg %d; refused sweeps %d; in the run: %d audited sweeps, %d refused, tail ratios %.2f / %.2f, min tau / estimate %.2f' % (
    c['sustained_steps'], c['sustained_ms_per_step_incl_recalibration'], c['sustained_meshes_per_s'], c['sustained_recalibrations'], c['sustained_refused_sweeps'],
    s['audited'], s['refused'], s['tail_ratio_max'], s.get('zoom_lattice_tail_ratio_max') or 0.0, s['min_tau_over_estimate']))" >> $O
done
python tools/cluster_soak.py 30000 2>&1 | tail -3 >> $O
cat $O
