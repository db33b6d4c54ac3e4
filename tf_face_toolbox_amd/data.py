"""Input pipeline: host-side mirror of the reference's data.py (Caffe-style `path label` lists,
random / class-balanced PxK sampling, decode -> [0,1] -> bilinear resize -> random crop -> flip or
augmentation -> (x-0.5)/0.5, batches in NHWC).  This is the caller side of the hot path (SURVEY.md
8f next-1): JPEG decode and augmentation stay on the host (PIL + numpy worker threads), a prefetch
thread keeps one batch ahead and hands over float32 NHWC CUDA tensors -- exactly the `inputs` dict
data.py:275-279 returns, with tensors replaced by callables that yield the next batch.

Reference: data.py:30-56 (list parsers), :77-96 (PxK dict), :153-191 (eval_inputs),
:195-281 (train_inputs)."""
import copy
import os
import sys
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from multiprocessing import cpu_count

import numpy as np
import torch

from .preprocessing import data_augmentation  # noqa: F401


def usable_cpus():
    """CPUs this process may actually use: the smaller of os.cpu_count(), the affinity mask and the cgroup CPU quota (a container
    limited to 16 cores of a 256-thread host reports 256).  Reported by bench.py / scripts/bench_loader.py next to their figures;
    the worker count is NOT derived from it: on such a host 48 decode workers still fed train.py 1.5x faster than 14."""
    n = cpu_count()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                    # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:                                                # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


# ------------------------------------------------------------------ list parsers (names kept)
def get_image_paths(list_path):
    """data.py:30-41: first whitespace-separated token of every line."""
    image_paths_flat = []
    for line in open(os.path.expanduser(list_path), 'r'):
        part_line = line.split()
        if part_line:
            image_paths_flat.append(part_line[0])
    return image_paths_flat, len(image_paths_flat)


def get_image_paths_and_labels(list_path):
    """data.py:44-58: `path label`, num_classes = max(label)+1."""
    image_paths_flat, labels_flat = [], []
    for line in open(os.path.expanduser(list_path), 'r'):
        part_line = line.split()
        if not part_line:
            continue
        image_paths_flat.append(part_line[0])
        labels_flat.append(int(part_line[1]))
    return image_paths_flat, labels_flat, len(labels_flat), max(labels_flat) + 1


def get_image_paths_and_labels_dict(list_path, num_per_class):
    """data.py:77-96: per-class path lists for the PxK sampler; classes with fewer than
    num_per_class images are dropped and the survivors are RE-NUMBERED 0..P'-1.  (The reference
    appends a single list when a new label shows up, so labels must first appear in increasing
    order there -- data.py:84-86; this version grows the table as far as needed instead.)"""
    image_path_list_tmp = []
    for line in open(os.path.expanduser(list_path), 'r'):
        part_line = line.split()
        if not part_line:
            continue
        label_index = int(part_line[1])
        while len(image_path_list_tmp) <= label_index:
            image_path_list_tmp.append([])
        image_path_list_tmp[label_index].append(part_line[0])
    image_path_list = [lst for lst in image_path_list_tmp if len(lst) >= num_per_class]
    return image_path_list, sum(len(lst) for lst in image_path_list), len(image_path_list)


# ------------------------------------------------------------------ decode / preprocess (host)
from ._decode_worker import resize_bilinear_tf1, decode as _decode, train_example as _train_example, fill_rows as _fill_rows   # noqa: E402,F401


class _Prefetcher(object):
    """One batch ahead on a background thread; `advance()` returns (images, labels) on `device`.
    A failure inside the producer (unreadable or corrupt image, malformed list line, label out of range) is caught,
    handed over through the queue and RE-RAISED from advance() in the training thread -- the reference's tf.data
    pipeline surfaces such errors to sess.run; a silently dead producer would leave this rank blocked on q.get() and
    the other ranks hanging in the all-reduce."""

    POLL_SECONDS = 5.0

    def __init__(self, make_batch, device, depth=2, num_classes=None, pool=None, transform=None):
        self.q = queue.Queue(maxsize=depth)
        self.transform = transform                                # raw uint8 slots on the device -> the float32 batch (fte_preprocess_u8)
        self.pool = pool                                          # a _WorkerPool with page-locked buffers, or None
        self.device = device
        self.make_batch = make_batch
        self.num_classes = num_classes
        self._cur = None
        self._dead = None
        # Host staging buffers, allocated ONCE and used in rotation (pinned when the device is a GPU): copying a 77 MB batch
        # into a fresh allocation every step cost 60-90 ms on the MI355X host (page faults of the new pages, 128 OpenMP threads
        # woken for one memcpy; `pin_memory()` registers new host memory every time) against 3 ms into a resident buffer --
        # it capped the pipeline at 5.5 k images/s whatever the number of decode workers.  depth + 3 buffers: the queue, the
        # batch the consumer holds, the one being filled, and one spare; a GPU copy's event is waited for before reuse.
        self._ring, self._slot = [], 0
        self._nslots = depth + 3
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _put(self, item):
        """q.put that gives up when close() has been called (the consumer is gone)"""
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.2)
                return True
            except queue.Full:
                pass
        return False

    def close(self, timeout=30.0):
        """Orderly shutdown (train.py / evaluate.py call it before they leave): stop the producer thread, wait for it, then end
        the decode workers and release their buffers.  Returns True when the thread has ended."""
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self._t.join(timeout)
        ended = not self._t.is_alive()
        if self.device.type == 'cuda':
            try:
                torch.cuda.synchronize()                      # host-to-device copies still reading the ring / the workers' buffers
            except Exception as e:                            # after a device fault: the callers' finally blocks must still run to their end
                print('input pipeline: device synchronise failed during close: %s' % e, file=sys.stderr)
                ended = False
        if self.pool is not None and ended:
            self.pool.close()
        if self._dead is None:
            self._dead = RuntimeError('input pipeline closed')
        return ended

    def _stage(self, x):
        """x (numpy, possibly the workers' shared buffer) -> a staging tensor of this prefetcher's ring"""
        if not self._ring or tuple(self._ring[0]['buf'].shape) != tuple(x.shape):
            self._ring = []
            for _ in range(self._nslots):
                b = torch.from_numpy(np.empty(x.shape, dtype=x.dtype))
                self._ring.append({'buf': b.pin_memory() if self.device.type == 'cuda' else b, 'ev': None})
            self._slot = 0
        e = self._ring[self._slot]
        self._slot = (self._slot + 1) % self._nslots
        if e['ev'] is not None:
            e['ev'].synchronize()                             # the host-to-device copy that read this buffer has finished
            e['ev'] = None
        np.copyto(e['buf'].numpy(), x, casting='same_kind')
        return e

    def _run(self):
        try:
            while not self._stop.is_set():
                x, y = self.make_batch()
                if y is not None and self.num_classes is not None and y.size:
                    lo, hi = int(y.min()), int(y.max())
                    if lo < 0 or hi >= self.num_classes:      # the loss kernels index rows / columns by label
                        raise ValueError('label out of range: batch has labels in [%d, %d], num_classes = %d' % (lo, hi, self.num_classes))
                yt = torch.from_numpy(y) if y is not None else None
                d = self.pool.direct(x) if (self.pool is not None and self.device.type == 'cuda') else None
                if d is not None:                                 # page-locked worker buffer: no staging copy at all
                    self._put(({'buf': d[0], 'ev': None, 'release': d[1]}, yt))
                else:
                    self._put((self._stage(x if isinstance(x, np.ndarray) else np.asarray(x, dtype=np.float32)), yt))
        except BaseException as e:                            # noqa: B902 -- everything goes to the consumer
            self._put(e)

    def advance(self):
        if self._dead is not None:
            raise self._dead
        while True:
            try:
                item = self.q.get(timeout=self.POLL_SECONDS)
                break
            except queue.Empty:
                if not self._t.is_alive():
                    self._dead = RuntimeError('input pipeline thread died without reporting an error')
                    raise self._dead
        if isinstance(item, BaseException):
            self._dead = RuntimeError('input pipeline failed: %s: %s' % (type(item).__name__, item))
            raise self._dead from item
        slot, yt = item
        if self.device.type == 'cuda':
            xd = slot['buf'].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if slot.get('release') is not None:
                slot['release'](ev)                           # worker buffer: the pool waits for it before the next refill
            else:
                slot['ev'] = ev                               # the producer waits for it before it overwrites the buffer
            if self.transform is not None:
                xd = self.transform(xd)
        else:
            xd = slot['buf'].clone()                          # CPU consumers (tests) may keep a batch: hand out a copy, not the ring buffer
        self._cur = (xd, yt.to(self.device, non_blocking=True) if yt is not None else None)
        return self._cur


class _WorkerPool(object):
    """Decode workers as separate PROCESSES (`python -m tf_face_toolbox_amd._decode_worker`: numpy + PIL only, no torch, no GPU)
    that write their rows straight into a float32 batch buffer in anonymous shared memory (memfd) mapped by everybody.  Threads top out at a
    few hundred images/s (the numpy part of decode/resize holds the GIL); the GPU step consumes 10-30 k images/s."""

    GROUP = 16       # workers that share one batch: with 128 workers on one 512-image batch each has ~5 ms of work per batch and is
    # asleep most of the time (wake-ups dominate) -- so the pool is cut into groups and every group works on a DIFFERENT batch.

    def __init__(self, workers, shape, pin=False, dtype=np.float32):
        import atexit
        import mmap
        import subprocess
        import sys
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.pin = bool(pin)
        self.procs, self.fds, self.mms, self.maps, self.tensors = [], [], [], [], None
        self.pin_failed = False
        atexit.register(self.close)                # registered before anything below can raise: a half-built pool is torn down too
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1',
                   OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        gs = min(self.GROUP, workers)
        ngroups = max(1, workers // gs)
        # Every group has TWO batches in the pipe (one being decoded, the next already waiting on its workers' stdin): a group
        # that has to wait for the parent to copy its batch out before it gets the next one idles a third of the time.
        self.DEPTH = 2 * ngroups
        # batch buffers in rotation: the open tickets + the one being copied out; with `pin` the buffers themselves are
        # page-locked (hipHostRegister on the shared mapping) and the GPU copies straight out of them, so a buffer also stays
        # busy while it waits in the prefetch queue and until its copy's event has fired: 4 more
        self.RING = self.DEPTH + (5 if self.pin else 1)
        # The batch buffers are ANONYMOUS shared memory (memfd_create), inherited by the workers as file descriptors: nothing has a
        # name under /dev/shm, so a rank that is SIGTERMed / SIGKILLed (torch.distributed.run does that to the survivors of a
        # failed rank; so do schedulers and `timeout`) leaves nothing behind -- the kernel frees the pages with the last process
        # that holds the descriptor -- and the size of the /dev/shm mount (64 MB by default in containers) does not matter.
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        for k in range(self.RING):
            try:
                fd = os.memfd_create('fte_batch_%d' % k)
            except (AttributeError, OSError):          # no memfd_create (non-Linux / old kernel / seccomp): an UNLINKED temporary file --
                import tempfile                        # just as nameless once it exists, freed with the last descriptor, but backed by
                tfd, tpath = tempfile.mkstemp(prefix='fte_batch_')      # the temporary directory's file system instead of anonymous memory
                os.unlink(tpath)
                fd = tfd
            self.fds.append(fd)
            os.ftruncate(fd, nbytes)
            mm = mmap.mmap(fd, nbytes)
            self.mms.append(mm)
            self.maps.append(np.frombuffer(mm, dtype=self.dtype).reshape(self.shape))
        self.procs = [subprocess.Popen([sys.executable, '-m', 'tf_face_toolbox_amd._decode_worker'], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, env=env, pass_fds=self.fds) for _ in range(workers)]
        self.groups = [self.procs[i:i + gs] for i in range(0, workers - gs + 1, gs)]
        if workers % gs:
            self.groups[-1] = self.groups[-1] + self.procs[workers - workers % gs:]
        self.turn = 0
        self.events = [None] * self.RING         # per buffer: the event of the host-to-device copy that last read it
        if self.pin:
            # page-lock the shared buffers so that the GPU copies straight out of them; if the runtime refuses (locked-memory
            # limit, exotic mapping) the batches go through _Prefetcher's pinned staging ring instead -- slower, never fatal
            rt = torch.cuda.cudart()
            tensors = [torch.from_numpy(m) for m in self.maps]
            done = []
            for t in tensors:
                if int(rt.cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)) != 0 or not t.is_pinned():
                    break
                done.append(t)
            if len(done) == len(tensors):
                self.tensors = tensors
            else:
                for t in done:
                    rt.cudaHostUnregister(t.data_ptr())
                self.pin = False
                self.pin_failed = True             # (train_inputs() passes it on as inputs['pin_fallback']: benchmarks can notice)
                print('tf_face_toolbox_amd.data: hipHostRegister of the shared batch buffers failed -- batches are staged through '
                      'pinned host buffers instead (a large loader slowdown)', file=sys.stderr)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=2)
            except Exception:
                p.kill()
        self.procs = []
        if self.tensors:
            try:
                torch.cuda.synchronize()
                rt = torch.cuda.cudart()
                for t in self.tensors:
                    rt.cudaHostUnregister(t.data_ptr())
            except Exception:
                pass
            self.tensors = None
        self.maps = []                            # the mmaps are released with the last numpy view of them (handed-out batches may outlive the pool)
        self.mms = []
        for fd in self.fds:
            try:
                os.close(fd)
            except OSError:
                pass
        self.fds = []

    def submit(self, rows, params):
        """Hand rows [(row, path, seed)] to the next group of workers; returns a ticket for wait().  At most DEPTH tickets may
        be open (a group works through its tasks in order), and they must be waited for in submission order."""
        import pickle
        import struct
        k = self.turn % self.RING
        procs = self.groups[self.turn % len(self.groups)]
        self.turn += 1
        if self.events[k] is not None:           # the GPU copy that read this buffer must have finished before workers refill it
            self.events[k].synchronize()
            self.events[k] = None
        n = len(procs)
        used = []
        for i, p in enumerate(procs):
            ch = rows[i::n]
            if not ch:
                continue
            b = pickle.dumps((self.fds[k], self.shape, ch) + tuple(params))
            p.stdin.write(struct.pack('<I', len(b)) + b)
            p.stdin.flush()
            used.append(p)
        return k, used

    def wait(self, ticket):
        """The filled batch array of a ticket (valid until RING - 1 further submits); worker errors are raised here."""
        import pickle
        import struct
        k, used = ticket
        err = None
        for p in used:
            hdr = p.stdout.read(4)
            if len(hdr) < 4:
                raise RuntimeError('a decode worker died (exit code %s)' % p.poll())
            kind, val = pickle.loads(p.stdout.read(struct.unpack('<I', hdr)[0]))
            if kind != 'ok' and err is None:
                err = val
        if err is not None:
            raise RuntimeError('decode worker: %s' % err)
        return self.maps[k]

    def direct(self, array):
        """(page-locked tensor over `array`, callback taking the copy's event) if `array` is one of this pool's registered
        buffers, else None"""
        if not self.tensors:
            return None
        for k, m in enumerate(self.maps):
            if m is array:
                return self.tensors[k], (lambda ev, k=k: self.events.__setitem__(k, ev))
        return None

    def fill(self, rows, params):
        return self.wait(self.submit(rows, params))


class _BatchSource(object):
    """images() advances to the next batch; labels() returns the labels of that same batch
    (one `sess.run` fetches both from one dataset element in the reference)."""

    def __init__(self, prefetcher):
        self.p = prefetcher

    def images(self):
        return self.p.advance()[0]

    def labels(self):
        if self.p._cur is None:
            self.p.advance()
        return self.p._cur[1]


def _raw_slots(num_workers, rows, num_channels, in_h, in_w, out_h, out_w, augmentation, device):
    """(slot bytes, transform) when the decode workers hand over DECODED uint8 images and the resize / crop / flip / normalise
    runs on the GPU (fte_preprocess_u8: the same bits as the host transform, tests/test_gpu_loader.py), else (0, None).  The
    host transform is 0.6 of the 1.7 ms a 250 x 250 JPEG costs a worker; the decode stays.  FTE_LOADER_GPU=0 keeps everything on
    the host; the colour augmentation (preprocessing.py) always does.  A slot holds an image of FTE_LOADER_RAW_SIDE^2 pixels
    (default 256; CASIA-WebFace crops are 250 x 250) -- a larger image is transformed by its worker and handed over finished."""
    if num_workers <= 0 or augmentation or torch.device(device).type != 'cuda' or os.environ.get('FTE_LOADER_GPU', '1') == '0':
        return 0, None
    from . import _lib
    from ._decode_worker import HEADER_BYTES
    side = int(os.environ.get('FTE_LOADER_RAW_SIDE', '256'))
    slot = HEADER_BYTES + max(side * side * num_channels, out_h * out_w * num_channels * 4)
    slot = (slot + 63) // 64 * 64

    def transform(raw):
        out = torch.empty((rows, out_h, out_w, num_channels), dtype=torch.float32, device=raw.device)
        _lib.call('fte_preprocess_u8', raw.data_ptr(), out.data_ptr(), rows, slot, num_channels, in_h, in_w, out_h, out_w,
                  torch.cuda.current_stream().cuda_stream)
        return out
    return slot, transform


# ------------------------------------------------------------------ public input builders
def train_inputs(data_list_path, input_height, input_width, crop_height=-1, crop_width=-1, is_color=1,
                 augmentation=0, batch_size=-1, num_classes=-1, num_per_class=-1, device='cuda', seed=None,
                 rank=0, world_size=1, num_workers=None):
    """data.py:195-281.  Returns {'images', 'labels', 'num_classes', 'num_examples', 'batch_size'};
    images/labels are callables (next batch / its labels).  With world_size > 1 every rank draws the
    same global batch order (shared seed) and decodes only its own rows [r*B/n, (r+1)*B/n)."""
    num_channels = 3 if is_color else 1
    rng = np.random.default_rng(seed)
    if batch_size == -1:
        assert num_classes != -1 and num_per_class != -1
        batch_size = num_classes * num_per_class
        image_label_dict, num_examples_total, num_classes_total = get_image_paths_and_labels_dict(data_list_path, num_per_class)

        def _gen():                                                  # data.py:230-242 (_gen_balance)
            work = copy.deepcopy(image_label_dict)
            for lst in work:
                rng.shuffle(lst)
            while True:
                for idx in rng.choice(len(work), num_classes):
                    if len(work[idx]) < num_per_class:
                        work[idx] = copy.deepcopy(image_label_dict[idx])
                        rng.shuffle(work[idx])
                    for _ in range(num_per_class):
                        yield work[idx].pop(), idx
    else:
        image_list, label_list, num_examples_total, num_classes_total = get_image_paths_and_labels(data_list_path)

        def _gen():                                                  # data.py:225-228 (_gen_random) + repeat()
            while True:
                for idx in rng.permutation(num_examples_total):
                    yield image_list[idx], label_list[idx]
    print('%d images loaded, totally %d classes' % (num_examples_total, num_classes_total))
    assert batch_size % world_size == 0
    shard = batch_size // world_size
    gen = _gen()
    out_h = crop_height if crop_height != -1 and crop_width != -1 else input_height
    out_w = crop_width if crop_height != -1 and crop_width != -1 else input_width
    # num_workers: decode PROCESSES (data.py:260 maps with cpu_count()/2 parallel calls); 0 = threads in this process
    # (small batches, tests).  Default: processes once a rank's shard is big enough to pay for starting them.
    if num_workers is None:
        num_workers = int(os.environ.get('FTE_LOADER_WORKERS', '-1'))
        if num_workers < 0:
            # measured on the MI355X host (to the GPU straight out of the page-locked worker buffers): 16 workers 18.7 k,
            # 32 workers 23.8 k, 48 workers 27.1 k images/s (through a staging copy: 24.5 k at 48)
            num_workers = min(max(1, cpu_count() // 2 // max(1, world_size)), 48, shard // 4) if shard >= 64 else 0
    on_gpu = torch.device(device).type == 'cuda'
    slot, transform = _raw_slots(num_workers, shard, num_channels, input_height, input_width, out_h, out_w, augmentation, device)
    if slot:
        procs = _WorkerPool(num_workers, (shard, slot), pin=on_gpu, dtype=np.uint8)
    else:
        procs = _WorkerPool(num_workers, (shard, out_h, out_w, num_channels), pin=on_gpu) if num_workers > 0 else None
    pool = None if procs else ThreadPoolExecutor(max(1, cpu_count() // 2))
    params = (num_channels, input_height, input_width, crop_height, crop_width, augmentation) + ((1,) if slot else ())

    def draw():
        items = [next(gen) for _ in range(batch_size)][rank * shard:(rank + 1) * shard]
        seeds = rng.integers(0, 2 ** 31, size=batch_size)[rank * shard:(rank + 1) * shard]
        return items, seeds, np.asarray([lab for _, lab in items], dtype=np.int32)

    pending = []                                         # worker processes: batch k + 1 is being decoded while batch k is copied out

    def make_batch():
        if procs is not None:
            while len(pending) < procs.DEPTH:
                items, seeds, labels = draw()
                pending.append((procs.submit([(i, it[0], int(sd)) for i, (it, sd) in enumerate(zip(items, seeds))], params), labels))
            ticket, labels = pending.pop(0)
            return procs.wait(ticket), labels
        items, seeds, labels = draw()
        imgs = list(pool.map(lambda a: _train_example(a[0][0], num_channels, input_height, input_width, crop_height,
                                                      crop_width, augmentation, np.random.default_rng(a[1])),
                             zip(items, seeds)))
        x = np.stack(imgs).reshape(shard, out_h, out_w, num_channels)
        return x, labels

    pf = _Prefetcher(make_batch, torch.device(device), num_classes=num_classes_total, pool=procs, transform=transform)
    src = _BatchSource(pf)

    def close():
        ended = pf.close()
        if pool is not None:
            pool.shutdown(wait=True)
        return ended
    return {'images': src.images, 'labels': src.labels, 'num_classes': num_classes_total,
            'num_examples': num_examples_total, 'batch_size': batch_size, 'close': close,
            'pin_fallback': bool(procs is not None and procs.pin_failed), 'gpu_transform': bool(slot)}


def eval_inputs(data_list_path, batch_size, is_color, input_height, input_width, device='cuda', num_workers=None):
    """data.py:153-191: repeating, in list order, resized, (x-0.5)/0.5.  Returns (next_batch, num_examples).  Batches of 64 or
    more are decoded by the worker processes train_inputs uses (threads in this process manage ~0.3 k images/s)."""
    num_channels = 3 if is_color else 1
    image_list, num_examples = get_image_paths(data_list_path)
    print('%d images loaded' % num_examples)
    if num_workers is None:
        num_workers = int(os.environ.get('FTE_LOADER_WORKERS', '-1'))
        if num_workers < 0:
            num_workers = min(max(1, cpu_count() // 2), 32, batch_size // 4) if batch_size >= 64 else 0
    on_gpu = torch.device(device).type == 'cuda'
    slot, transform = _raw_slots(num_workers, batch_size, num_channels, input_height, input_width, input_height, input_width, 0, device)
    if slot:
        procs = _WorkerPool(num_workers, (batch_size, slot), pin=on_gpu, dtype=np.uint8)
    else:
        procs = _WorkerPool(num_workers, (batch_size, input_height, input_width, num_channels), pin=on_gpu) if num_workers > 0 else None
    pool = None if procs else ThreadPoolExecutor(8)
    state = {'pos': 0}
    params = (num_channels, input_height, input_width, -1, -1, 0) + ((1,) if slot else ())
    pending = []

    def draw():
        paths = [image_list[(state['pos'] + i) % num_examples] for i in range(batch_size)]
        state['pos'] = (state['pos'] + batch_size) % num_examples
        return paths

    def make_batch():
        if procs is not None:
            while len(pending) < procs.DEPTH:
                pending.append(procs.submit([(i, q, None) for i, q in enumerate(draw())], params))
            return procs.wait(pending.pop(0)), None
        imgs = list(pool.map(lambda q: (_decode(q, num_channels, input_height, input_width) - 0.5) / 0.5, draw()))
        return np.stack(imgs).astype(np.float32), None

    pf = _Prefetcher(make_batch, torch.device(device), pool=procs, transform=transform)

    def next_batch():
        return pf.advance()[0]

    def close():
        ended = pf.close()
        if pool is not None:
            pool.shutdown(wait=True)
        return ended
    next_batch.close = close                     # orderly shutdown of the producer thread and the decode workers (evaluate.py)
    return next_batch, num_examples


def synthetic_inputs(batch_size, height, width, is_color, num_classes, device='cuda', rank=0, world_size=1, seed=0):
    """Not in the reference: a resident random batch (bench.py's inputs) for runs without a list file."""
    ch = 3 if is_color else 1
    g = torch.Generator().manual_seed(seed)
    shard = batch_size // world_size
    x = (torch.rand(batch_size, height, width, ch, generator=g) * 2 - 1)[rank * shard:(rank + 1) * shard].to(device)
    y = torch.randint(0, num_classes, (batch_size,), generator=g, dtype=torch.int32)[rank * shard:(rank + 1) * shard].to(device)
    return {'images': x, 'labels': y, 'num_classes': num_classes, 'num_examples': batch_size * 100, 'batch_size': batch_size}
