"""CPU: bench.py's launcher logic (no GPU here, so every rank fails early -- which is exactly what is checked)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_gpus_n_without_a_distributed_environment_spawns_its_own_ranks():
    """`python bench.py --gpus 2` must not die with "launch with torch.distributed.run" (round-1 behaviour): it starts the
    ranks itself.  Without a GPU the ranks fail in torch.cuda.set_device; the parent reports that and exits non-zero
    without printing a JSON line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0
    assert 'the 2-rank run failed' in r.stderr
    assert 'launch with torch.distributed.run' not in r.stderr + r.stdout
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_world_size_mismatch_is_refused():
    env = _env()
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and '--gpus 4 but WORLD_SIZE=2' in r.stderr


def test_kernel_source_stamp_is_stable_and_traffic_file_is_found():
    sys.path.insert(0, ROOT)
    import bench
    a, b = bench.kernel_src_sha(), bench.kernel_src_sha()
    assert a == b and len(a) == 16
    p = bench.latest_traffic_file()
    assert p is None or os.path.basename(p).endswith('_traffic.json')


def test_library_carries_the_hash_of_the_kernel_sources():
    """csrc/build.sh stamps libfte.so with the hash of csrc/*.hip, *.h; bench.py prints the LIBRARY's stamp as kernel_src_sha and
    refuses to run when it differs from the sources' (the tie between binary and sources was mtime only)."""
    sys.path.insert(0, ROOT)
    import bench
    from tf_face_toolbox_amd import _lib
    assert _lib.version().endswith('src:' + bench.kernel_src_sha()), (_lib.version(), bench.kernel_src_sha())
    assert bench.library_src_sha() == bench.kernel_src_sha()
    assert bench.check_library_stamp() == bench.kernel_src_sha()


def test_other_configs_is_the_last_key_of_the_line():
    """the other BASELINE.json configs (and SphereNet's 2 / 4 / 8-GPU shards) are timed on the driver's clock and
    reported under `other_configs`, the LAST key of the N = 1 line (the driver keeps the tail).  Checked on the source: the last
    assignment into `out` before the print, and the list itself."""
    sys.path.insert(0, ROOT)
    import bench
    names = [(n, d, b) for n, d, b in bench.OTHER_CONFIGS]
    assert ('ResNeXt-50-center', 'bf16s', 128) in names and ('SENet-50-triplet', 'bf16s', 128) in names
    assert ('ShuffleNet-v2-small', 'f32', 256) in names
    assert [b for n, d, b in names if n == 'SphereNet-ASoftmax' and d == 'f32'] == [256, 128, 64]
    src = open(os.path.join(ROOT, 'bench.py')).read()
    head, _ = src.rsplit('print(json.dumps(out))', 1)
    assigns = [ln.strip() for ln in head.splitlines() if ln.strip().startswith("out['")]
    assert assigns[-1].startswith("out['other_configs'] = time_other_configs("), assigns[-3:]
    from tf_face_toolbox_amd.nets.net_base import net_select      # every name is a factory name
    for n, _, _ in names:
        assert net_select(n, 'NCHW', 5e-4) is not None
