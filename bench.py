#!/usr/bin/env python
"""bench.py -- images/sec of the data-parallel training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work: started WITHOUT a torch.distributed environment and with --gpus N > 1, the script launches its own N
ranks (child processes through torch.distributed.run, before anything in this process touches the GPU), relays rank
0's JSON line and exits with the children's status.

Workload (BASELINE.json configs[1]): SphereFaceNet-20 + A-softmax, 112x112 RGB,
GLOBAL batch 512 (train.py --batch_size is the global batch; data_parallel.py:206 splits it),
C = 10,575 classes, fp32, Momentum 0.9, wd 5e-4, lr 1e-4 (at the reference default 0.1 this synthetic
task -- one random-label batch fitted over and over by a BN-free net -- diverges to inf within 4-15
steps; lr only scales the update, the work per step is identical, and the run must stay finite).  A step = forward + loss + backward +
gradient all-reduce + optimizer (one `sess.run(train_ops)` of train.py:228) on synthetic inputs
already resident in HBM.  Strong scaling: rank r works on rows [r*512/N, (r+1)*512/N).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel symbol (most device time among the MFMA launches; since round 6 the Winograd product of the
                 conv3x3 forward pass, wino_mm_kernel<0,2>; with FTE_CONV_ALGO=direct an igemm_kernel<...> instantiation).  Its
                 launches inside the recorded steps (two of the timed steps at 512 images, every 4th where the net walks one
                 stream anyway) are bracketed by a HIP event pair on the launch stream (fte_prof_*, include/fte.h); a recorded
                 step runs ONE chain of kernels (no second stream, no half shards) so that a launch's duration is its own;
                 achieved = sum of the FLOPs the launches EXECUTE (Winograd: 2*16*tiles*N*K; direct: 2*rows*N*K) / sum of
                 their durations, against the 157.3 TFLOP/s fp32 matrix peak.  avg_launch_ms is directly comparable with the
                 AverageNs of the same symbol in profiles/*kernel_stats.csv (collected with --one-stream: every step that way).  `per_shape` lists every (op, GEMM shape) of the step: fwd / dgrad / wgrad of
                 the four stages, the stride-2 entries, FC and classifier -- ms, TFLOP/s, fraction of peak, and the
                 algorithmic bytes / s of that launch (every operand and result tensor once);
  cpu_baseline : BASELINE.md section 3: the float32 torch-CPU restatement of the reference graph (oracle/torch_ref.py,
                 kind "port") timed on this host's cores at configs[0] -- 64 gray 112x112 images, Singular path, whole
                 training steps -- with CPU model + thread count, and `parity`: max-abs / rel-L2 of the HIP path's
                 embeddings and logits against it on those same 64 images (rank 0, N=1 only);
  allreduce    : N > 1: RCCL world size, bucket sizes, each bucket's all-reduce timed alone after the run, and the step
                 time with the all-reduce switched off (what the collective costs after overlap).
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GLOBAL_BATCH = 512
H = W = 112
CH = 3
NUM_CLASSES = 10575
FP32_MFMA_PEAK_TFLOPS = 157.3                    # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0                   # same table, "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBPS = 8000.0                           # same table, HBM3E peak (spec)
LR = 1e-4
TILES = {0: '128,128', 1: '256,64', 2: '128,64', 3: '64,64', 4: '192,64', 5: '64,64 (1 wave)', 6: '64,64 (2 waves)',   # igemm.h's TILE_* enum
         7: 'resident filter gradient (wgrad16.hip): 9 taps x 32|64 cin x 256|128 cout, or a 1x1 channel tile',
         8: 'Winograd F(2x2,3x3) / F(3x3,2x2) (wino.hip): 64 x 64 x 16 planes per resident block'}
CONV_ALGOS = {0: 'direct', 1: 'winograd', 2: 'auto'}
# (name, MFMA dtype, images) of the other BASELINE.json configs' per-GPU shards and of SphereNet's 2 / 4 / 8-GPU shards: timed after the
# headline region and the CPU baseline, reported under `other_configs` (the LAST key of the line)
OTHER_CONFIGS = [('ResNeXt-50-center', 'bf16s', 128), ('SENet-50-triplet', 'bf16s', 128), ('ShuffleNet-v2-small', 'f32', 256),
                 ('SphereNet-ASoftmax', 'f32', 256), ('SphereNet-ASoftmax', 'f32', 128), ('SphereNet-ASoftmax', 'f32', 64)]


def kernel_src_sha():
    """Identity of the kernel sources: the committed PMC traffic figures are only valid for the kernels they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'tf_face_toolbox_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'fte.h'), 'rb').read())      # the ABI header is part of the build (csrc/build.sh)
    return h.hexdigest()[:16]


def library_src_sha():
    """The hash the LOADED libfte.so was built from (csrc/build.sh stamps it into fte_version())."""
    from tf_face_toolbox_amd import _lib
    return _lib.source_stamp()


def check_library_stamp():
    """A stale libfte.so must not report numbers under the current sources' name: refuse to run (FTE_BENCH_ALLOW_STALE=1 overrides,
    for A/B variants loaded through FTE_LIB)."""
    lib, src = library_src_sha(), kernel_src_sha()
    if lib != src and os.environ.get('FTE_BENCH_ALLOW_STALE') != '1' and not os.environ.get('FTE_LIB'):
        raise SystemExit('bench.py: libfte.so was built from other kernel sources (library src:%s, csrc/ is %s): run '
                         'tf_face_toolbox_amd/csrc/build.sh (python -c "import __graft_entry__ as g; g.build()")' % (lib or 'unstamped', src))
    return lib


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_and_parity(dev, mfma_dtype='f32', min_seconds=10.0, max_steps=12):
    """configs[0] = SphereFaceNet-20 + A-softmax, 112x112 GRAY, batch 64, one replica.
    Times whole training steps of the float32 torch-CPU restatement (oracle/torch_ref.py) on the host cores, then
    runs the HIP path on the same 64 images / same weights and reports the parity of embeddings and logits."""
    import numpy as np
    import torch
    from oracle import torch_ref, spherenet as osn, ops as oops
    from tf_face_toolbox_amd import net_select
    n, ch = 64, 1
    p = osn.init_params(2, ch, NUM_CLASSES, H, W, dtype=np.float32)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (n, H, W, ch)).astype(np.float32)
    y = np.random.default_rng(1).integers(0, NUM_CLASSES, n)
    lam = float(oops.asoftmax_lambda(0))
    threads = torch.get_num_threads()
    from tf_face_toolbox_amd.data import usable_cpus          # affinity mask and cgroup CPU quota
    quota = usable_cpus()
    # ---- parity first (weights untouched): CPU forward vs HIP forward on the same inputs ----
    tp = torch_ref.to_torch(p, torch.float32, requires_grad=False)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    with torch.no_grad():
        _, _, emb_c, log_c = torch_ref.spherenet_loss(tp, xt, yt, 5e-4, 'NCHW', 'asoftmax', lam)
    net = net_select('SphereNet-ASoftmax', 'NCHW', 5e-4)
    net.build(H, W, ch, NUM_CLASSES, dev)
    net.load_params(p)
    net.global_step = 0
    out = net.forward(xt.to(dev), yt.to(dev, torch.int32), num_classes=NUM_CLASSES, is_training=True)
    emb_g = net.emb.float().cpu()
    log_g = out['logits'].float().cpu()
    torch.cuda.synchronize()

    def maxabs(a, b):
        return float((a - b).abs().max())

    def rell2(a, b):
        return float(((a - b).double().pow(2).sum() / b.double().pow(2).sum().clamp_min(1e-300)).sqrt())
    tol = 1e-4 if mfma_dtype == 'f32' else 2e-2          # bf16 MFMA operands: the stated mixed-precision tolerance
    parity = {'config': 'configs[0]: 64 gray 112x112 images, SphereFaceNet-20 + A-softmax (lambda %.0f), same weights' % lam,
              'embedding_maxabs': maxabs(emb_g, emb_c), 'embedding_max_ref': float(emb_c.abs().max()),
              'embedding_rell2': rell2(emb_g, emb_c),
              'logits_maxabs': maxabs(log_g, log_c), 'logits_max_ref': float(log_c.abs().max()),
              'logits_rell2': rell2(log_g, log_c),
              'tolerance': 'max-abs <= %g * max|ref| (%s HIP path vs fp32 CPU path)' % (tol, {'f32': 'fp32', 'bf16': 'bf16-operand', 'bf16s': 'bf16-storage'}[mfma_dtype])}
    parity['ok'] = bool(parity['embedding_maxabs'] <= tol * parity['embedding_max_ref'] and
                        parity['logits_maxabs'] <= tol * parity['logits_max_ref'])
    del net
    # ---- timing: whole training steps on the host cores ----
    tp = torch_ref.to_torch(p, torch.float32, requires_grad=True)
    slots = {k: torch.zeros_like(v) for k, v in tp.items()}
    torch_ref.train_step(tp, slots, xt, yt, LR, 5e-4, 'NCHW', 'asoftmax', lam)          # untimed: thread pools, oneDNN primitives
    # a fair baseline uses the thread count the CPU library runs this batch fastest at (64 images do not scale to every
    # core of a 2-socket host): one untimed step per candidate, the best one is timed
    best = (None, threads)
    for cand in sorted({threads, max(1, threads // 2), max(1, threads // 4), max(1, threads // 8), min(threads, quota)}, reverse=True):
        torch.set_num_threads(cand)
        t0 = time.time()
        torch_ref.train_step(tp, slots, xt, yt, LR, 5e-4, 'NCHW', 'asoftmax', lam)
        dt = time.time() - t0
        if best[0] is None or dt < best[0]:
            best = (dt, cand)
    threads = best[1]
    torch.set_num_threads(threads)
    t0 = time.time()
    reps = 0
    while True:
        torch_ref.train_step(tp, slots, xt, yt, LR, 5e-4, 'NCHW', 'asoftmax', lam)
        reps += 1
        el = time.time() - t0
        if el >= min_seconds or reps >= max_steps:
            break
    base = {'value': round(n * reps / el, 3), 'unit': 'images/sec', 'cores': int(threads), 'kind': 'port',
            'cpu_model': cpu_model(), 'os_cpu_count': os.cpu_count(), 'usable_cpus': quota,
            'sample': '%d training steps (fwd + A-softmax loss + bwd + Momentum) of BASELINE configs[0] = 64 gray 112x112 images, '
                      'float32 torch-CPU restatement of the reference graph (oracle/torch_ref.py: F.conv2d + autograd), %.1f s on %d threads'
                      % (reps, el, threads)}
    return base, parity


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as FRESH child processes (nothing in
    this process has touched the GPU -- it never will), relay rank 0's JSON line, exit non-zero if any rank failed."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + '\n')
    if r.returncode != 0 or line is None:
        sys.stderr.write('bench.py: the %d-rank run failed (exit code %d)\n' % (n, r.returncode))
        raise SystemExit(r.returncode or 1)
    print(line)
    sys.stdout.flush()
    raise SystemExit(0)


def latest_traffic_file(dtype='f32'):
    """profiles/r<N>_traffic.json (fp32 run) or profiles/r<N>_bf16_traffic.json of the highest round N."""
    import re
    d = os.path.join(ROOT, 'profiles')
    pat = re.compile(r'^r(\d+)_traffic\.json$' if dtype == 'f32' else r'^r(\d+)_%s_traffic\.json$' % dtype)
    best = None
    for f in os.listdir(d) if os.path.isdir(d) else []:
        m = pat.match(f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(d, f))
    return best[1] if best else None


def op_kind(key, sym=''):
    if sym.startswith('wino_mm_kernel'):          # wino_mm_kernel<EPI, channel halves per block>
        return 'dgrad' if sym.startswith('wino_mm_kernel<1') else 'fwd'
    if sym.startswith('wino_wgrad'):
        return 'wgrad'
    al, bl, epi = key[:3]
    if al == 1:
        return 'wgrad'                           # A = x^T (k = pixel): filter gradient / dense tn
    if epi == 0 and bl == 1 and (sym.startswith('igemm16') or sym.endswith(',2>')):
        return 'fwd'                             # bf16-source forward: the weights are packed [tap][cout][cin] (the NK layout)
    return 'dgrad' if epi == 1 or bl == 1 else 'fwd'


def time_other_configs(dev, steps0=10, warmup0=3):
    """One GPU, synthetic inputs, whole training steps (forward + loss + backward + optimizer) of the other BASELINE.json configs at their
    per-GPU shards and of SphereNet at the 2 / 4 / 8-GPU shards: device time over `steps0` steps after `warmup0` (four times as many for
    shards of <= 128 images), by events on the stream."""
    import torch
    from tf_face_toolbox_amd import net_select, Singular, _lib
    out = []
    prev = _lib.precision_mode()
    for name, mode, b in OTHER_CONFIGS:
        try:
            _lib.set_mfma_dtype(mode)
            g = torch.Generator().manual_seed(7)
            x = (torch.rand(b, H, W, CH, generator=g) * 2 - 1).to(dev)
            y = torch.randint(0, NUM_CLASSES, (b,), generator=g, dtype=torch.int32).to(dev)
            net = net_select(name, 'NCHW', 5e-4)
            step, losses, names, _ = Singular(net, 1e-4, 'Momentum')({'images': x, 'labels': y, 'num_classes': NUM_CLASSES, 'num_examples': b,
                                                                     'batch_size': b})
            # (the short steps get more of them: 13 steps of 6 ms end before the clocks have settled -- 64 images read 6.56 ms here
            # against 5.97 ms in a 120-step run of the same shard)
            steps, warmup = (steps0, warmup0) if b > 128 else (4 * steps0, 4 * warmup0)
            for _ in range(warmup):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps
            dev_ms = e0.elapsed_time(e1) / steps
            vals = [float(l) for l in losses]
            out.append({'net': name, 'dtype': mode, 'images': b, 'ms_per_step': round(1e3 * wall, 3), 'device_ms_per_step': round(dev_ms, 3),
                        'images_per_sec': round(b / wall, 1), 'steps': steps, 'finite': all(v == v and abs(v) < 1e30 for v in vals)})
            del step, net, x, y, losses
            gc.collect()                           # (the net and its step closure are a reference cycle: left to the collector, its buffers are
            torch.cuda.empty_cache()               # freed -- device-synchronising hipFree calls -- inside the NEXT config's timed steps)
        except Exception as e:                     # a config that fails is reported, never silently dropped
            out.append({'net': name, 'dtype': mode, 'images': b, 'error': '%s: %s' % (type(e).__name__, e)})
    _lib.set_mfma_dtype(prev)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)      # mean over >= 50 timed steps after >= 10 warm-up
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the `other_configs` leg (N = 1 only, ~15 s)')
    ap.add_argument('--one-stream', action='store_true',
                    help='profiling only: EVERY step runs as the launch-record steps do (one chain of kernels, no second stream), so that '
                         'rocprofv3 kernel durations and PMC bytes are each launch\'s own (scripts/collect_profiles.sh); the value printed is '
                         'then not the metric')
    ap.add_argument('--global-batch', type=int, default=GLOBAL_BATCH,
                    help='exploration only (e.g. the per-rank shard sizes of N=2/4/8 on one GPU); the metric is quoted at 512')
    ap.add_argument('--mfma-dtype', choices=['f32', 'bf16', 'bf16s'], default='f32',
                    help="operand precision of the MFMA products (fte_set_mfma_dtype).  The metric (BASELINE.json configs[1]) is "
                         "fp32 = the default; bf16 = bf16 operands, fp32 accumulate, fp32 storage (exploration, configs[2]'s precision)")
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])     # does not return

    import torch
    import torch.distributed as dist
    from tf_face_toolbox_amd import net_select, Singular, DataParallel_margin

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    # FTE_BENCH_SHARED_GPU=1 (tests only): all ranks on the GPUs that exist, gloo as the transport -- RCCL refuses two
    # ranks on one device, and the test boxes have one GPU; everything else of the N > 1 path is what the driver runs
    shared = os.environ.get('FTE_BENCH_SHARED_GPU') == '1'
    if shared:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'gloo' if shared else 'nccl'
        if shared:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    gb = args.global_batch
    assert gb % world == 0
    shard = gb // world

    # synthetic inputs: images U[-1,1] seed 0, labels seed 1, reference initialisers seed 2
    g = torch.Generator().manual_seed(0)
    images = (torch.rand(gb, H, W, CH, generator=g) * 2 - 1)[rank * shard:(rank + 1) * shard].to(dev)
    g = torch.Generator().manual_seed(1)
    labels = torch.randint(0, NUM_CLASSES, (gb,), generator=g, dtype=torch.int32)[rank * shard:(rank + 1) * shard].to(dev)
    from tf_face_toolbox_amd import _lib
    check_library_stamp()                         # libfte.so is the build of THESE sources, or no numbers
    _lib.set_mfma_dtype(args.mfma_dtype)          # before the wrapper's construction pass: every launch of this process runs in this mode
    net = net_select('SphereNet-ASoftmax', 'NCHW', 5e-4)
    net.seed = 2
    inputs = {'images': images, 'labels': labels, 'num_classes': NUM_CLASSES, 'num_examples': 494414,
              'batch_size': gb}
    if world > 1:
        model = DataParallel_margin(net, LR, 'Momentum', num_gpus=world, weight_decay=5e-4)
    else:
        model = Singular(net, LR, 'Momentum', weight_decay=5e-4)
    train_ops, losses, losses_name, others = model(inputs)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    peak = FP32_MFMA_PEAK_TFLOPS if args.mfma_dtype == 'f32' else BF16_MFMA_PEAK_TFLOPS
    net.one_stream = args.one_stream
    for _ in range(args.warmup):
        train_ops()
    barrier()
    # Launch records (event pairs around every MFMA-kernel launch, for the roofline leg) are taken on every PROF_EVERY-th
    # timed step: an event pair costs a queue barrier per launch -- recording all steps lowers the measured rate by 4.6 %
    # at 64 images per GPU (7.57 k -> 7.24 k images/s), by 0.5 % at 512.  FTE_BENCH_NO_PROF=1 turns them off (exploration).
    PROF_EVERY = 4
    two_streams = getattr(net, '_side_stream', None) is not None and net._side_stream(shard) is not None
    if two_streams:
        # small shards / bf16 mode: the backward walk uses two streams and a recorded step does not (below) -- two recorded steps
        # (the first and the middle one) are enough for the per-shape table and cost the timed region < 1 %
        PROF_EVERY = max(4, (args.steps + 1) // 2)
    prof = os.environ.get('FTE_BENCH_NO_PROF') != '1'
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step device times (no host sync)
    net.one_stream = args.one_stream
    first = True
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if prof and i % PROF_EVERY == 0:
            _lib.query('fte_prof_enable', 1 if first else 2)
            first = False
            net.one_stream = True          # the recorded steps run the one-stream backward walk: a launch's duration is then its own,
                                           # not that of two kernels sharing the chip (small shards / bf16 mode: nets/sphere.py _side_stream)
        train_ops()
        if prof and i % PROF_EVERY == 0:
            _lib.query('fte_prof_enable', 0)
            net.one_stream = args.one_stream
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    _lib.query('fte_prof_enable', 0)
    records = _lib.prof_records(shapes=True) if rank == 0 else []
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_vals = [float(l) for l in losses]
    import math
    if not all(math.isfinite(v) for v in loss_vals):
        raise SystemExit('non-finite losses %s: the timed run is invalid' % loss_vals)

    # ---- N > 1: what the collective costs (outside the timed region) ----
    allreduce = None
    if world > 1:
        buckets = net.grad_buckets()
        alone = []
        for a, b in buckets:
            buf = torch.zeros(b - a, dtype=torch.float32, device=dev)
            for _ in range(2):
                dist.all_reduce(buf)
            barrier()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                dist.all_reduce(buf)
            barrier()
            alone.append(1000.0 * (time.perf_counter() - t1) / reps)
            del buf

        class _NoComm(object):                   # same wrapper, collective switched off: the step's compute-only time
            def __init__(self, c): self.c = c
            def world_size(self): return self.c.world_size()
            def rank(self): return self.c.rank()
            def broadcast(self, t, src=0): pass

            def all_reduce_async(self, t):
                class _W(object):
                    def wait(self): return True
                return _W()
        real = model.comm
        model.comm = _NoComm(real)
        ksteps = max(2, args.steps // 2)
        for _ in range(2):
            train_ops()
        barrier()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            train_ops()
        barrier()
        nocomm = torch.tensor([1000.0 * (time.perf_counter() - t1) / ksteps], dtype=torch.float64, device=dev)
        dist.all_reduce(nocomm, op=dist.ReduceOp.MAX)
        model.comm = real
        allreduce = {'backend': backend + (' (RCCL over xGMI)' if backend == 'nccl' else ' (test transport)'),
                     'rccl_ranks': dist.get_world_size() if backend == 'nccl' else 0,
                     'bucket_bytes': [4 * (b - a) for a, b in buckets],
                     'bucket_alone_ms': [round(v, 3) for v in alone],
                     'bucket_busbw_GBps': [round(4 * (b - a) * 2 * (world - 1) / world / (ms_ * 1e-3) / 1e9, 1)
                                           for (a, b), ms_ in zip(buckets, alone)],
                     'ms_per_step_without_allreduce': round(float(nocomm.item()), 3)}

    if rank == 0:
        ms = 1000.0 * elapsed / args.steps
        if not records:                      # FTE_BENCH_NO_PROF=1: throughput only
            print(json.dumps({'value': round(gb * args.steps / elapsed, 2), 'ms_per_step': round(ms, 3), 'n_gpus': world, 'note': 'launch records off'}))
            return
        sampled = (args.steps + PROF_EVERY - 1) // PROF_EVERY
        # per-symbol table: key = the kernel symbol the launch was dispatched to, as the library recorded it
        # (fte_prof_get_name: the template arguments rocprofv3 prints) -> [launches, flops, ms, alg bytes]; kinds: its op
        table, shapes, kinds = {}, {}, {}
        for sig, fl, ms_, mnk, by, sym in records:
            t = table.setdefault(sym, [0, 0.0, 0.0, 0.0])
            kinds[sym] = op_kind(sig, sym)
            t[0] += 1; t[1] += fl; t[2] += ms_; t[3] += by
            s = shapes.setdefault((op_kind(sig, sym),) + tuple(mnk) + (sig[3], sig[4]), [0, 0.0, 0.0, 0.0])
            s[0] += 1; s[1] += fl; s[2] += ms_; s[3] += by
        DOM = max(table, key=lambda k: table[k][2])              # the symbol with the most device time
        cnt, dom_flops, dom_ms, dom_bytes = table[DOM]
        avg_ms = dom_ms / cnt
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        all_ms = sum(v[2] for v in table.values())
        all_flops = sum(v[1] for v in table.values())
        # PMC traffic: cannot be collected inside this process; measured by rocprofv3 --pmc on this same command and kept
        # under profiles/, stamped with the kernel sources it was measured on (stale figures are nulled, not reported)
        traffic, tsrc, tnote = None, None, None
        tpath = latest_traffic_file(args.mfma_dtype)
        if world == 1 and tpath and gb == GLOBAL_BATCH:
            tinfo = json.load(open(tpath))
            ent = (tinfo.get('symbols') or {}).get(DOM)
            if tinfo.get('kernel_src_sha') != library_src_sha():
                tnote = 'profiles/%s was measured on other kernel sources (%s != library %s): traffic nulled' % (
                    os.path.basename(tpath), tinfo.get('kernel_src_sha'), library_src_sha())
            elif ent and tinfo.get('mfma_dtype', 'f32') == args.mfma_dtype:
                traffic = ent['hbm_bytes_per_launch']
                tsrc = 'profiles/' + os.path.basename(tpath)
        per_shape = []
        for key in sorted(shapes, key=lambda k: -shapes[k][2]):
            c_, fl_, ms_, by_ = shapes[key]
            tf_ = fl_ / (ms_ * 1e-3) / 1e12
            per_shape.append({'op': key[0], 'rows': key[1], 'N': key[2], 'K': key[3], 'tile': TILES.get(key[4], '?'), 'splits': key[5],
                              'launches_per_step': round(c_ / sampled, 2), 'ms': round(ms_ / c_, 4),
                              'tflops': round(tf_, 1), 'frac': round(tf_ / peak, 3),
                              'alg_GBps': round(by_ / (ms_ * 1e-3) / 1e9, 0)})
        out = {
            'metric': 'images/sec (whole node), SphereFaceNet-20 112x112 bs512',
            'value': round(gb * args.steps / elapsed, 2),
            'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3),
            'step_ms': {'min': round(step_ms[0], 3), 'median': round(step_ms[len(step_ms) // 2], 3), 'max': round(step_ms[-1], 3)},
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': args.mfma_dtype,
            'data': 'synthetic',
            'config': {'workload': 'SphereFaceNet-20 + A-softmax training step, 112x112x3, global batch %d, 10575 classes, Momentum, %s' % (gb, {'f32': 'fp32', 'bf16': 'bf16 MFMA operands / fp32 accumulate + storage', 'bf16s': 'bf16 MFMA operands + bf16 storage of activations and inter-layer gradients / fp32 accumulate, sums, master weights'}[args.mfma_dtype]),
                       'global_batch': gb, 'per_gpu_batch': shard, 'lr': LR, 'parallelism': 'dp%d' % world,
                       'train_gflop_per_image': 12.2698},
            # the algorithm of the stride-1 3x3 layers (fte.h FTE_CONV_*; the reference runs Winograd: train.py:260) and how many launches of
            # a recorded step took it
            'conv_algo': {'setting': CONV_ALGOS.get(_lib.query('fte_get_conv_algo'), '?'),
                          'winograd_launches_per_step': round(sum(v[0] for k, v in table.items() if k.startswith('wino_')) / sampled, 2),
                          'direct_mfma_launches_per_step': round(sum(v[0] for k, v in table.items() if not k.startswith('wino_')) / sampled, 2),
                          'note': 'Winograd F(2x2,3x3) forward / data gradient and F(3x3,2x2) filter gradient on the stride-1 3x3 layers of '
                                  '>= 128 channels, forward + filter gradient of the 64-channel stage; direct implicit GEMM elsewhere',
                          # how the timed steps walk the net (the launch-record steps behind `roofline` run one chain of kernels instead)
                          'forward_walk': ('two half shards on two streams' if getattr(net, '_fwd_halves', None) is not None and not args.one_stream
                                           and net._fwd_halves(shard, _lib.get_mfma_dtype() == 'bf16') else 'one chain'),
                          'backward_walk': 'filter gradients on a second stream' if two_streams and not args.one_stream else 'one chain'},
            # step_mfma_frac: DIRECT-CONVOLUTION-EQUIVALENT FLOPs of the step (12.27 GFLOP per image) / time / peak -- the
            # metric's own FLOP count; under Winograd the kernels execute fewer FLOPs, so this figure is not bounded by 1.
            # step_mfma_frac_executed: the FLOPs the MFMA launches of a recorded step actually execute / the step time / peak (<= 1)
            'step_mfma_frac': round(gb * args.steps / elapsed * 12.2698e9 / (peak * 1e12) / world, 4),
            'step_mfma_frac_basis': 'direct-convolution-equivalent FLOPs (12.27 GFLOP per image)',
            'step_mfma_frac_executed': round(all_flops / sampled / (elapsed / args.steps) / (peak * 1e12), 4),
            'losses': dict(zip(losses_name, [round(v, 6) for v in loss_vals])),
            'roofline': {'bound': 'mfma',
                         'kernel': '%s = %s (%s)' % (DOM, {'fwd': 'conv3x3 forward + bias/PReLU/residual', 'dgrad': 'conv3x3 data gradient + PReLU gradient', 'wgrad': 'conv3x3 filter gradient'}[kinds[DOM]],
                                                       'Winograd: 16 plane products on the fp32 MFMA; FLOPs = those the launch EXECUTES' if DOM.startswith('wino_') else 'MFMA implicit GEMM'),
                         'achieved': round(achieved, 2), 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                         'launches_timed': cnt, 'avg_launch_ms': round(avg_ms, 4),
                         'flops_per_launch_avg': dom_flops / cnt,
                         'traffic': traffic, 'traffic_source': tsrc, 'traffic_note': tnote,
                         'algorithmic_bytes_per_launch': round(dom_bytes / cnt),
                         'algorithmic_GBps': round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1),
                         'all_mfma_kernels': {'launches': len(records), 'steps_sampled': sampled,
                                              'ms_per_step': round(all_ms / sampled, 3),
                                              'achieved': round(all_flops / (all_ms * 1e-3) / 1e12, 2),
                                              'frac': round(all_flops / (all_ms * 1e-3) / 1e12 / peak, 4)},
                         'per_symbol': {k: {'op': kinds[k], 'launches_per_step': round(v[0] / sampled, 2), 'ms_per_step': round(v[2] / sampled, 3),
                                                             'tflops': round(v[1] / (v[2] * 1e-3) / 1e12, 1),
                                                             'frac': round(v[1] / (v[2] * 1e-3) / 1e12 / peak, 3),
                                                             'algorithmic_bytes_per_launch': round(v[3] / v[0])}
                                        for k, v in sorted(table.items(), key=lambda kv: -kv[1][2])},
                         'per_shape': per_shape},
            'kernel_src_sha': library_src_sha(),          # the LIBRARY's stamp (= the sources': check_library_stamp())
        }
        if allreduce is not None:
            out['allreduce'] = allreduce
        if world == 1 and not args.no_cpu_baseline:
            del train_ops, model
            out['cpu_baseline'], out['parity'] = cpu_baseline_and_parity(dev, args.mfma_dtype)
        else:
            out['cpu_baseline'] = None
        if world == 1 and not args.no_other_configs and gb == GLOBAL_BATCH and args.mfma_dtype == 'f32':
            train_ops = model = None
            del net, images, labels, inputs
            torch.cuda.empty_cache()
            out['other_configs'] = time_other_configs(dev)       # LAST key: it lands in the tail of the line
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
