"""gpurun_out/r06_reference_cpu_on_gpu_box.jsonl (one line per tools/time_reference.py run of tools/gpu_jobs/with_reference_job.sh)
-> profiles/r06_reference_cpu_on_gpu_box.json: the box fingerprint, every run, and per configuration the Serial figure (north_star's
"serial-vec CPU path") and the best figure over all backends / thread counts tried.  bench.py reads this file."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, 'gpurun_out', 'r06_reference_cpu_on_gpu_box.jsonl')
runs = [json.loads(l) for l in open(src) if l.strip().startswith('{')]
assert runs, 'no runs'
box = {k: runs[0][k] for k in ('cpu_model', 'cores_physical', 'cores_logical', 'torch')}
box['date'] = runs[0].get('date')
assert all(all(r[k] == box[k] for k in box if k != 'date') for r in runs), 'runs from different boxes'
summary = {}
for cfg in sorted({r['config'] for r in runs}):
    rs = [r for r in runs if r['config'] == cfg]
    serial = max((r for r in rs if r['backend'] == 'serial'), key=lambda r: r['value'])
    best = max(rs, key=lambda r: r['value'])
    short = lambda r: {k: r[k] for k in ('value', 'unit', 'backend', 'workers', 'torch_threads', 'cores_used', 'envs', 'horizon', 'iterations',  # noqa: E731
                                         'evaluate_s_per_iter', 'train_s_per_iter', 'profile', 'what')}
    summary[cfg] = dict(serial=short(serial), best=short(best))
doc = dict(what='the UNMODIFIED reference (PufferLib 1.0.1: clean_pufferl.create/evaluate/train, pufferlib.vector.Serial / Multiprocessing, '
                'c_gae.pyx via pyximport) timed on the GPU box\'s own host cores by tools/gpu_jobs/with_reference.sh (the reference files '
                'travel in a git-ignored staging directory that is removed after the job); BASELINE.md section 3 configurations',
           box=box, summary=summary, runs=runs)
out = os.path.join(REPO, 'profiles', 'r06_reference_cpu_on_gpu_box.json')
json.dump(doc, open(out, 'w'), indent=1)
print(out)
for cfg, s in summary.items():
    print(cfg, 'serial %.1f k (threads %d)' % (s['serial']['value'] / 1e3, s['serial']['torch_threads']),
          '| best %.1f k (%s, workers %s, threads %d)' % (s['best']['value'] / 1e3, s['best']['backend'], s['best']['workers'], s['best']['torch_threads']))
