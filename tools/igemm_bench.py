"""Per-product timing of the NatureCNN's implicit-GEMM launches at one 8192-frame chunk (developer tool, not product code).

    python tools/igemm_bench.py build  name1:-DFLAG=1  name2:...     (here, no GPU: hipcc cross-compiles csrc/igemm.hip per variant)
    python tools/igemm_bench.py run [frames]                         (on the GPU box; with no variants built: the product library's two
                                                                      product forms, fp32 MFMA and the six-term bf16 split, side by side)

Every product of cnn.Engine.forward / backward (conv1..3 and the Linear: forward, dX, dW) is launched through the C-ABI entry points
the engine uses, timed with HIP events on the stream they run on (median of 5 after 2 warm-ups), and summarised by a checksum of its
output so that a variant that computes something else shows up next to its time.  TFLOP/s = algorithmic flop (2 m n k) / time."""
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tools', '_probe')


def build(specs):
    from pufferlib_amd import _lib
    _lib.build()
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, 'libig_*.so')):
        os.remove(f)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(':')
        flags = [f for f in flags.split(',') if f]
        obj = os.path.join(OUT, f'igemm_{name}.o')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', *flags,
               os.path.join(_lib.CSRC, 'igemm.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [os.path.join(_lib.LIB_DIR, os.path.splitext(s)[0] + '.o') for s in _lib.SOURCES if s != 'igemm.hip'] + [obj]
        so = os.path.join(OUT, f'libig_{name}.so')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + objs + ['-ldl'])
        os.remove(obj)
        print(so)


def one(so, n, products=0, dump=''):
    import numpy as np
    import torch
    from pufferlib_amd import _lib
    if so:
        _lib.LIB_PATH = so
    from pufferlib_amd import cnn, models
    _lib.check(_lib.lib().pfa_igemm_set_products(int(products)), 'set_products')

    class _Env:
        single_action_space = type('Discrete', (), {'n': 4})()
    torch.manual_seed(0)
    net = models.Convolutional(_Env(), framestack=4, flat_size=64 * 7 * 7)
    cp = models.ConvParams(net, 'cuda')
    eng = cnn.Engine(cp, chunk=n)
    eng.pack()
    g = torch.Generator(device='cuda').manual_seed(1)
    frames = torch.randint(0, 256, (n, 4 * 84 * 84), dtype=torch.uint8, device='cuda', generator=g)
    eng.forward(frames, n)
    dh = torch.randn(n, 512, device='cuda', generator=g) * (eng.h > 0)
    grads = torch.zeros(cp.count + 16, device='cuda')
    gv = cp.split(grads[:cp.count])
    c1, c2, c3, fc = eng.conv1, eng.conv2, eng.conv3, eng.fc
    products = [
        ('conv1 forward', lambda: c1.forward(frames, n, eng.a1), 2 * n * 400 * 32 * 256, lambda: eng.a1),
        ('conv2 forward', lambda: c2.forward(eng.a1, n, eng.a2), 2 * n * 81 * 64 * 512, lambda: eng.a2),
        ('conv3 forward', lambda: c3.forward(eng.a2, n, eng.a3), 2 * n * 49 * 64 * 576, lambda: eng.a3),
        ('linear forward', lambda: fc.forward(eng.a3, n, eng.h), 2 * n * 512 * 3136, lambda: eng.h),
        ('linear dW', lambda: fc.backward_dw(eng.a3, n, dh, gv['network.7.weight'], gv['network.7.bias'], False, eng.ws), 2 * n * 512 * 3136,
         lambda: gv['network.7.weight']),
        ('linear dX', lambda: fc.backward_dx(dh, n, eng.a3, eng.d3), 2 * n * 512 * 3136, lambda: eng.d3),
        ('conv3 dW', lambda: c3.backward_dw(eng.a2, n, eng.d3, gv['network.4.weight'], gv['network.4.bias'], False, eng.ws), 2 * n * 49 * 64 * 576,
         lambda: gv['network.4.weight']),
        ('conv3 dX', lambda: c3.backward_dx(eng.d3, n, eng.a2, eng.d2), 2 * n * 49 * 64 * 576, lambda: eng.d2),
        ('conv2 dW', lambda: c2.backward_dw(eng.a1, n, eng.d2, gv['network.2.weight'], gv['network.2.bias'], False, eng.ws), 2 * n * 81 * 64 * 512,
         lambda: gv['network.2.weight']),
        ('conv2 dX', lambda: c2.backward_dx(eng.d2, n, eng.a1, eng.d1), 2 * n * 81 * 64 * 512, lambda: eng.d1),
        ('conv1 dW', lambda: c1.backward_dw(frames, n, eng.d1, gv['network.0.weight'], gv['network.0.bias'], False, eng.ws), 2 * n * 400 * 32 * 256,
         lambda: gv['network.0.weight']),
    ]
    res, samples = {}, {}
    product_list, products = products, None
    for name, fn, flop, out in product_list:
        for _ in range(2):
            fn()
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        us = sorted(times)[2]
        o = out().double()
        res[name] = dict(us=round(us, 1), tflops=round(flop / us / 1e6, 1), sum=float(o.sum()), abs=float(o.abs().sum()))
        samples[name] = out().reshape(-1)[:1 << 20].float().cpu().numpy()
    if dump:
        np.savez(dump, **{k.replace(' ', '_'): v for k, v in samples.items()})
    print(json.dumps(res))


def run(n):
    import numpy as np
    libs = sorted(glob.glob(os.path.join(OUT, 'libig_*.so')))
    jobs = [(os.path.basename(so)[6:-3], so, 0) for so in libs] or [('fp32', '', 0), ('bf16x6', '', 1)]   # no variants built: the product's two forms
    table = {}
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    for name, so, prod in jobs:
        dump = os.path.join(REPO, 'gpurun_out', f'igb_{name}.npz')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'one', so, str(n), str(prod), dump], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=280)
        if r.returncode != 0:
            print(name, 'FAILED', r.stderr.decode()[-600:])
            continue
        table[name] = json.loads(r.stdout.decode().strip().splitlines()[-1])
    names = list(table)
    if not names:
        return 1
    first = table[names[0]]
    print(f'{"product":16s}' + ''.join(f'{v:>22s}' for v in names))
    for prod in first:
        row = f'{prod:16s}'
        for v in names:
            t = table[v][prod]
            same = abs(t['sum'] - first[prod]['sum']) <= 1e-5 * max(1.0, first[prod]['abs'])
            row += f'{t["us"]:9.1f} us {t["tflops"]:6.1f} TF{"" if same else " !"}'
        print(row)
    tot = {v: sum(table[v][p]['us'] for p in first) for v in names}
    print(f'{"sum":16s}' + ''.join(f'{tot[v]:9.1f} us {"":9s}' for v in names))
    if len(names) > 1:     # element-wise distance of every variant's outputs (first 2^20 elements) from the first variant's
        base = np.load(os.path.join(REPO, 'gpurun_out', f'igb_{names[0]}.npz'))
        for v in names[1:]:
            other = np.load(os.path.join(REPO, 'gpurun_out', f'igb_{v}.npz'))
            print(f'max |{v} - {names[0]}| / max |{names[0]}| per product: ' +
                  ', '.join(f"{k.replace('_', ' ')} {np.abs(other[k] - base[k]).max() / max(np.abs(base[k]).max(), 1e-30):.1e}" for k in base.files))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'igemm_bench_last.json'), 'w') as f:
        json.dump(table, f, indent=1)
    return 0


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    elif sys.argv[1] == 'one':
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0, sys.argv[5] if len(sys.argv) > 5 else '')
    else:
        sys.exit(run(int(sys.argv[2]) if len(sys.argv) > 2 else 8192))
