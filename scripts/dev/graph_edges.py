"""HIP-graph replay against the eager step, kernel by kernel (find the edges).

  run (on the GPU box, under rocprofv3 --kernel-trace):   python3 scripts/dev/graph_edges.py run ResNeXt-50-center 128
  analyse the trace:                                        python3 scripts/dev/graph_edges.py cmp <kernel_trace.csv>

`run` executes 6 eager steps, then 6 replays of one captured step, each phase bracketed by a marker kernel (a cos_ on a 12345-element
tensor).  `cmp` splits the trace at the markers and prints, per phase: wall per step, sum of kernel durations, union of busy intervals,
time with two kernels in flight, idle time, the gap distribution between consecutive kernels, and the kernels whose average duration
differs most between the phases."""
import collections, csv, os, re, sys


def run(name, B):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    from tf_face_toolbox_amd import net_select, Singular
    ncls = 10575
    g = torch.Generator().manual_seed(0)
    hw = (112, 96) if name.startswith('SphereNet') else (112, 112)
    x = (torch.rand(B, hw[0], hw[1], 3, generator=g) * 2 - 1).cuda()
    y = torch.randint(0, ncls, (B,), generator=g, dtype=torch.int32).cuda()
    net = net_select(name, 'NCHW', 5e-4)
    step, losses, names, _ = Singular(net, 1e-3, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': B})
    mark = torch.zeros(12345, device='cuda')
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        step()
    torch.cuda.synchronize()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    mark.cos_()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    mark.cos_()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    mark.cos_()
    for _ in range(6):
        gr.replay()
    torch.cuda.synchronize()
    mark.cos_()
    torch.cuda.synchronize()
    print('done', flush=True)


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n)[:64]


def phase(seg, nsteps, label):
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
    wall = (max(e for _, e in iv) - iv[0][0]) / nsteps / 1e3
    busy = sum(e - s for s, e in iv) / nsteps / 1e3
    # sweep: time with >= 1 and >= 2 kernels in flight
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last, t1, t2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1: t1 += t - last
        if depth >= 2: t2 += t - last
        depth += d; last = t
    # gaps: from the end of everything so far to the next start
    gaps, hi = [], iv[0][1]
    for s, e in iv[1:]:
        if s > hi: gaps.append(s - hi)
        hi = max(hi, e)
    gaps.sort()
    q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3 if gaps else 0.0
    print('%-6s wall %.1f us/step, %d launches/step, kernel time %.1f, busy (union) %.1f, two in flight %.1f, idle %.1f; %d gaps/step: median %.2f us, p90 %.2f, max %.1f, sum %.1f'
          % (label, wall, len(seg) / nsteps, busy, t1 / nsteps / 1e3, t2 / nsteps / 1e3, wall - t1 / nsteps / 1e3, len(gaps) / nsteps,
             q(0.5), q(0.9), q(0.999), sum(gaps) / nsteps / 1e3))
    print('       streams / queues: %s' % sorted(set((r.get('Stream_Id', '?'), r.get('Queue_Id', '?')) for r in seg)))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        a = agg[short(r['Kernel_Name'])]
        a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return {k: (c / nsteps, d / nsteps / 1e3) for k, (c, d) in agg.items()}


def cmp(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'cos' in r['Kernel_Name'] and 'elementwise' in r['Kernel_Name']]
    assert len(marks) >= 4, 'markers: %d' % len(marks)
    m = marks[-4:]
    eager = phase(rows[m[0] + 1:m[1]], 6, 'eager')
    graph = phase(rows[m[2] + 1:m[3]], 6, 'graph')
    print('%-66s %7s %9s %9s %9s' % ('kernel', 'calls', 'eager us', 'graph us', 'delta'))
    keys = sorted(set(eager) | set(graph), key=lambda k: -abs(graph.get(k, (0, 0))[1] - eager.get(k, (0, 0))[1]))
    for k in keys[:30]:
        e, g = eager.get(k, (0, 0)), graph.get(k, (0, 0))
        print('%-66s %7.1f %9.1f %9.1f %+9.1f' % (k, max(e[0], g[0]), e[1], g[1], g[1] - e[1]))
    print('%-66s %7s %9.1f %9.1f %+9.1f' % ('all kernels', '', sum(v[1] for v in eager.values()), sum(v[1] for v in graph.values()),
                                          sum(v[1] for v in graph.values()) - sum(v[1] for v in eager.values())))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], int(sys.argv[3]))
    else:
        cmp(sys.argv[2])
