"""-m gpu: oracle parity for EVERY tile instantiation the headline run (SphereFaceNet-20, 512 x 112 x 112 x 3) launches.

Round-2 review: the planner (csrc/api.hip plan_rows / wgrad_plan) picks the 192x64 filter-gradient tile only for K ranges of
>= 1024 pixels per split (>= ~84 images of 56x56), the tall 128x64 forward / dgrad tile only at >= 4096 such tiles, and split counts
of 153-256 only at the full batch -- none of which the small cases of test_gpu_kernels.py reach, so those instantiations were only
ever compared with themselves (additivity 512 = 256 + 256).  Here each one runs at a size that triggers it NATURALLY (no hook) and
is compared with the float64 oracle block by block; small hooked cases add ragged shapes.  The launch records (fte_prof_get_name)
say which kernel symbol actually ran, and the last test checks the union against the symbol list of the committed bench line.
Shapes: nets/sphere.py:56-70 at 112x112."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SEEN = set()          # kernel symbols the cases of this module ran (filled in file order; the last test reads it)


def _run(cases, env=None, timeout=900):
    e = dict(os.environ)
    for k in ('FTE_WGRAD_TILE', 'FTE_NARROW_TILE', 'FTE_WIDE_TILE', 'FTE_WGRAD_SPLIT_MAJOR', 'FTE_SPLIT_MINK', 'FTE_SK', 'FTE_SK_TILE', 'FTE_SK_WS_FLAGS'):
        e.pop(k, None)
    e['FTE_CONV_ALGO'] = 'direct'          # this module pins the tile instantiations of the DIRECT family; the Winograd symbols have their own test below
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, 'tile_worker.py'), json.dumps(cases)], env=e, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert lines, 'tile_worker produced no result line:\n' + r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(lines[-1])
    for c in res['cases']:
        SEEN.update(c['symbols'])
        assert c['ok'], 'parity failure: %s' % json.dumps(c)
    assert r.returncode == 0, r.stderr[-4000:]
    return res['cases']


def _has(case, symbol):
    assert symbol in case['symbols'], '%s did not run for %s (ran: %s)' % (symbol, case['case'], case['symbols'])


def test_wgrad_192x64_natural_and_256_way_split():
    """stage 1 (56x56, 64 -> 64): M = 9*64 = 576 = three 192-row tiles; 96 images give K ranges >= 1024 pixels, the planner keeps the
    192x64 tile and splits K ~256 ways, split-major (one pixel range per XCD)."""
    c = _run([['wgrad', 96, 56, 56, 64, 64, 1]])[0]
    _has(c, 'igemm_kernel<192,64,2,2,1,0,0,0>')
    assert max(c['splits']) >= 200, c['splits']


def test_wgrad_128x128_natural_many_splits():
    """stages 2 and 3: 128x128 filter-gradient tile with ~85-170 splits (28x28x128->128: nine tiles), plus the stride-2 entry conv
    of stage 2 (56x56x64 -> 28x28x128, M = 576 on the 128x128 tile: 4.5 tiles, ragged M) and a 14x14x256 layer (42 splits in the headline)."""
    cs = _run([['wgrad', 384, 28, 28, 128, 128, 1], ['wgrad', 384, 56, 56, 64, 128, 2], ['wgrad', 384, 14, 14, 256, 256, 1]])
    for c in cs:
        _has(c, 'igemm_kernel<128,128,2,2,1,0,0,0>')
    assert max(cs[0]['splits']) >= 80 and max(cs[1]['splits']) >= 80, (cs[0]['splits'], cs[1]['splits'])


def test_tall_128x64_fwd_dgrad_n64_natural():
    """stage 1 at >= 4096 tiles of 128x64: forward and data gradient of 56x56x64->64 on the tall tile (N = 64)."""
    cs = _run([['fwd', 168, 56, 56, 64, 64, 1], ['dgrad', 168, 56, 56, 64, 64, 1]])
    _has(cs[0], 'igemm_kernel<128,64,2,2,0,0,0,0>')
    _has(cs[1], 'igemm_kernel<128,64,2,2,0,1,1,0>')


def test_tall_128x64_fwd_dgrad_n128_natural():
    """stage 2 at >= 4096 tiles: 28x28x128->128 forward / data gradient on the tall tile (N = 128: two column tiles), and the stage-2
    entry conv's forward (56x56x64 -> 28x28x128, stride 2, K = 576)."""
    cs = _run([['fwd', 336, 28, 28, 128, 128, 1], ['dgrad', 336, 28, 28, 128, 128, 1], ['fwd', 336, 56, 56, 64, 128, 2]])
    _has(cs[0], 'igemm_kernel<128,64,2,2,0,0,0,0>')
    _has(cs[1], 'igemm_kernel<128,64,2,2,0,1,1,0>')
    _has(cs[2], 'igemm_kernel<128,64,2,2,0,0,0,0>')


def test_default_64x64_fwd_dgrad_at_many_rounds():
    """stages 3 and 4 stay on the 64x64 tile; sizes with several rounds of blocks and a ragged last round, and the stride-2 data
    gradient's four parity classes as separate launches (>= one round of tiles per class)."""
    cs = _run([['fwd', 176, 14, 14, 256, 256, 1], ['dgrad', 176, 14, 14, 256, 256, 1], ['dgrad', 256, 28, 28, 128, 256, 2],
               ['fwd', 344, 7, 7, 512, 512, 1], ['dgrad', 344, 7, 7, 512, 512, 1]])      # (> 4 tiles of 128x64 per CU: below that, stream-K)
    _has(cs[0], 'igemm_kernel<64,64,2,2,0,0,0,0>')
    _has(cs[1], 'igemm_kernel<64,64,2,2,0,1,1,0>')
    _has(cs[2], 'igemm_kernel<64,64,2,2,0,1,1,0>')
    assert len(cs[2]['splits']) == 4, cs[2]           # four class launches
    _has(cs[3], 'igemm_kernel<64,64,2,2,0,0,0,0>')
    _has(cs[4], 'igemm_kernel<64,64,2,2,0,1,1,0>')


@pytest.mark.parametrize('split_major', ['1', '0'])
def test_wgrad_192x64_hooked_small_and_ragged(split_major):
    """FTE_WGRAD_TILE=4 forces the 192x64 tile on small shapes: ragged K (pixels not a multiple of 32), a ragged last split,
    stride 2, M = 1152 (six tiles), >= 100 splits; with split-major block placement (the default) and with the 2-D grid."""
    env = {'FTE_WGRAD_TILE': '4', 'FTE_WGRAD_SPLIT_MAJOR': split_major}
    cs = _run([['wgrad', 3, 9, 7, 64, 64, 1], ['wgrad', 40, 28, 28, 64, 64, 1], ['wgrad', 5, 13, 13, 64, 64, 2],
               ['wgrad', 9, 14, 14, 128, 64, 1], ['wgrad', 37, 28, 28, 64, 128, 1]], env)
    for c in cs:
        _has(c, 'igemm_kernel<192,64,2,2,1,0,0,0>')
    assert max(cs[1]['splits']) >= 100, cs[1]['splits']


def test_wgrad_128x128_hooked_ragged_splits():
    """the 128x128 and 128x64 filter-gradient tiles at > 100 splits with a ragged last split (FTE_SPLIT_MINK lowers the K floor)."""
    cs = _run([['wgrad', 21, 28, 28, 128, 128, 1], ['wgrad', 21, 27, 29, 128, 256, 2]], {'FTE_WGRAD_TILE': '0', 'FTE_SPLIT_MINK': '96'})
    for c in cs:
        _has(c, 'igemm_kernel<128,128,2,2,1,0,0,0>')
    assert max(cs[0]['splits']) >= 100, cs[0]['splits']
    cs = _run([['wgrad', 21, 28, 28, 64, 64, 1]], {'FTE_WGRAD_TILE': '2', 'FTE_SPLIT_MINK': '96'})
    _has(cs[0], 'igemm_kernel<128,64,2,2,1,0,0,0>')


def test_persistent_bf16_kernels_at_the_sizes_that_select_them():
    """bf16 storage (fte_conv2d_{fwd,dgrad}_s16): at >= 768 tiles of 128 rows the planner takes the 128x128 / 128x64 tile and the 3x3 /
    stride-1 launches run on igemm16rw_kernel -- resident blocks (loader waves + consumer waves) that walk several tiles, the A operand
    through a padded-slot window in LDS, swapped-operand MFMAs, the register epilogue with the half-wave exchange (256 x 128) or the
    row-coalesced epilogue through LDS (256 x 64), column partials carried across a block's tiles -- or, by hook, on igemm16p_kernel.  Every stored bf16 value must be within half a bf16 step (+ fp32
    noise) of the float64 result of the same bf16 inputs; dalpha / dbias as in the fp32 cases."""
    env = {'FTE_MFMA_DTYPE': 'bf16s'}
    cs = _run([['s16fwd', 64, 56, 56, 64, 64, 1], ['s16dgrad', 64, 56, 56, 64, 64, 1], ['s16fwd', 126, 28, 28, 128, 128, 1],
               ['s16dgrad', 126, 28, 28, 128, 128, 1], ['s16fwd', 262, 14, 14, 256, 256, 1]], env, timeout=1500)
    _has(cs[0], 'igemm16rw_kernel<256,64,8,1,0,4,2,56,1,8,0>')         # K = 576: one barrier per K-step; row-coalesced epilogue
    _has(cs[1], 'igemm16rw_kernel<256,64,8,1,1,4,2,56,1,8,0>')
    _has(cs[2], 'igemm16rw_kernel<256,128,4,2,0,4,1,45,2,0,0>')          # two K-steps per barrier
    _has(cs[3], 'igemm16rw_kernel<256,128,4,2,1,4,1,45,2,0,0>')
    _has(cs[4], 'igemm16rw_kernel<256,128,4,2,0,4,1,45,2,0,0>')
    # ... one barrier per K-step on the 256 x 128 tile as well (hook), and a 7x7x512 layer (the unsplit plan: 72 K-steps per tile)
    cs = _run([['s16fwd', 126, 28, 28, 128, 128, 1], ['s16dgrad', 126, 28, 28, 128, 128, 1]], dict(env, FTE_IGEMM16_PERSIST='22'), timeout=1500)
    _has(cs[0], 'igemm16rw_kernel<256,128,4,2,0,4,2,48,1,0,0>')
    _has(cs[1], 'igemm16rw_kernel<256,128,4,2,1,4,2,48,1,0,0>')
    cs = _run([['s16fwd', 336, 7, 7, 512, 512, 1], ['s16dgrad', 336, 7, 7, 512, 512, 1]], env, timeout=1500)
    _has(cs[0], 'igemm16rw_kernel<256,128,4,2,0,4,1,45,2,0,0>')
    _has(cs[1], 'igemm16rw_kernel<256,128,4,2,1,4,1,45,2,0,0>')
    # the register epilogue on the 256 x 64 tile (FTE_IGEMM16_STG=0) and the row-coalesced one through the B ring on 256 x 128 (=2)
    cs = _run([['s16fwd', 64, 56, 56, 64, 64, 1], ['s16dgrad', 64, 56, 56, 64, 64, 1]], dict(env, FTE_IGEMM16_STG='0'), timeout=1500)
    _has(cs[0], 'igemm16rw_kernel<256,64,4,2,0,4,2,56,1,0,0>')
    _has(cs[1], 'igemm16rw_kernel<256,64,4,2,1,4,2,56,1,0,0>')
    cs = _run([['s16fwd', 126, 28, 28, 128, 128, 1], ['s16dgrad', 126, 28, 28, 128, 128, 1]], dict(env, FTE_IGEMM16_STG='2'), timeout=1500)
    _has(cs[0], 'igemm16rw_kernel<256,128,4,2,0,4,2,45,2,16,0>')
    _has(cs[1], 'igemm16rw_kernel<256,128,4,2,1,4,2,45,2,16,0>')
    # the same layers on the persistent kernel without loader waves / window (FTE_IGEMM16_PERSIST=14: every eligible launch)
    cs = _run([['s16fwd', 64, 56, 56, 64, 64, 1], ['s16dgrad', 64, 56, 56, 64, 64, 1], ['s16fwd', 126, 28, 28, 128, 128, 1],
               ['s16dgrad', 126, 28, 28, 128, 128, 1]], dict(env, FTE_IGEMM16_PERSIST='14'), timeout=1500)
    _has(cs[0], 'igemm16p_kernel<128,64,4,2,0,2,6,1,0>')
    _has(cs[1], 'igemm16p_kernel<128,64,4,2,1,2,6,1,0>')
    _has(cs[2], 'igemm16p_kernel<128,128,4,2,0,2,4,1,0>')
    _has(cs[3], 'igemm16p_kernel<128,128,4,2,1,2,4,1,0>')


def test_resident_filter_gradient_of_the_bf16_storage_mode():
    """wgrad16.hip: all nine taps per block, the reduction over PADDED slots (zero rows between image rows / images), x and dz through
    LDS-DMA and the transposing LDS read, slot-range splits summed by the slab reduction -- against the float64 oracle on the same
    bf16 inputs, the three instantiations (32 cin x 256 cout: 14x14x256, 7x7x512; 64 x 128: 28x28x128; 64 x 64: 56x56x64), ragged last ranges, and a
    shape the plan leaves to the per-tile kernel (too few K-pieces per block)."""
    cs = _run([['s16wgrad', 80, 14, 14, 256, 256, 1], ['s16wgrad', 70, 7, 7, 512, 512, 1], ['s16wgrad', 83, 28, 28, 128, 128, 1],
               ['s16wgrad', 24, 14, 14, 256, 256, 1], ['s16wgrad', 45, 56, 56, 64, 64, 1], ['s16wgrad', 1, 7, 7, 256, 256, 1]],
              {'FTE_MFMA_DTYPE': 'bf16s'}, timeout=1500)
    _has(cs[4], 'wgrad16_kernel<64,64,3,192>')
    _has(cs[0], 'wgrad16_kernel<32,256,3,128>')
    _has(cs[1], 'wgrad16_kernel<32,256,3,128>')
    _has(cs[2], 'wgrad16_kernel<64,128,3,128>')
    _has(cs[3], 'wgrad16_kernel<32,256,3,128>')          # round 5: fewer slot ranges (S lowered to 8 K-pieces per block) instead of the per-tile plan
    assert not any(s.startswith('wgrad16') for s in cs[5]['symbols']), cs[5]['symbols']      # one K-piece in all: the per-tile kernel


def test_resident_filter_gradient_at_the_bn_nets_128_image_shard():
    """Round 5: the 3x3 layers of ResNet-50 / SE-ResNet-50 at 128 images -- 28x28x64, 14x14x128, 7x7x256
    (nets/resnet.py:63-92) -- have 225-1000 K-pieces, fewer than eight per block at one block per CU: wgrad16_plan lowers the number of
    slot ranges instead of handing them to the register-staged 64x64 kernel (44-49 us -> 30-44 us per layer)."""
    cs = _run([['s16wgrad', 128, 28, 28, 64, 64, 1], ['s16wgrad', 128, 14, 14, 128, 128, 1], ['s16wgrad', 128, 7, 7, 256, 256, 1]],
              {'FTE_MFMA_DTYPE': 'bf16s'}, timeout=1500)
    _has(cs[0], 'wgrad16_kernel<64,64,3,192>')
    _has(cs[1], 'wgrad16_kernel<64,128,3,128>')
    _has(cs[2], 'wgrad16_kernel<32,256,3,128>')


def test_pointwise_resident_filter_gradient():
    """wgrad16p_kernel (1x1 convs of the bf16 storage mode): 64-pixel K-pieces of x and dz through LDS-DMA and the transposing read, pixel
    ranges summed by the slab reduction -- the three channel tiles (128 x 256, 256 x 128, 128 x 128), a ragged last K-piece (pixel count
    no multiple of 64), few pixels (fewer ranges than CUs), and channel counts it leaves to the per-tile kernel."""
    cs = _run([['s16wgrad1', 37, 28, 28, 128, 256, 1], ['s16wgrad1', 128, 14, 14, 512, 128, 1], ['s16wgrad1', 9, 27, 29, 128, 128, 1],
               ['s16wgrad1', 12, 7, 7, 1024, 2048, 1], ['s16wgrad1', 16, 14, 14, 64, 64, 1]], {'FTE_MFMA_DTYPE': 'bf16s'}, timeout=1500)
    _has(cs[0], 'wgrad16p_kernel<128,256,2,4>')
    _has(cs[1], 'wgrad16p_kernel<256,128,4,2>')
    _has(cs[2], 'wgrad16p_kernel<128,128,2,4>')
    _has(cs[3], 'wgrad16p_kernel<128,256,2,4>')
    assert not any(s.startswith('wgrad16p') for s in cs[4]['symbols']), cs[4]['symbols']
    # ... and the tiles with a 64-channel side (128-byte rows: the other swizzle key)
    cs = _run([['s16wgrad1', 21, 28, 28, 64, 256, 1], ['s16wgrad1', 21, 28, 28, 256, 64, 1], ['s16wgrad1', 33, 14, 14, 64, 128, 1],
               ['s16wgrad1', 33, 13, 15, 128, 64, 1]], {'FTE_MFMA_DTYPE': 'bf16s'}, timeout=1500)
    _has(cs[0], 'wgrad16p_kernel<64,256,1,8>')
    _has(cs[1], 'wgrad16p_kernel<256,64,8,1>')
    _has(cs[2], 'wgrad16p_kernel<64,128,2,4>')
    _has(cs[3], 'wgrad16p_kernel<128,64,4,2>')


def test_stream_k_natural_at_the_small_shards():
    """Round 5: launches of < 4 tiles (128x64) per CU run on igemm_sk_kernel (csrc/igemm.hip "stream-K"; planner: api.hip plan_sk) --
    forward on 64x64 tiles, data gradient on 128x64.  The 8-GPU shard of the headline (64 images) NATURALLY: 14x14x256 (1.53 tiles per
    CU, every tile split over 2-3 workers), 7x7x512 (0.78 per CU: was split-K + fix-up), 28x28x128 (3.06), the stride-2 entry of
    stage 3 (forward); the 128-image shard's 14x14 layer; and 56x56x64 must stay on the one-block-per-tile kernel (6 tiles per CU).
    reference: /root/reference/data_parallel.py:206-207 (the shard), nets/sphere.py:61-70 (the layers)."""
    cs = _run([['fwd', 64, 14, 14, 256, 256, 1], ['dgrad', 64, 14, 14, 256, 256, 1], ['fwd', 64, 7, 7, 512, 512, 1], ['dgrad', 64, 7, 7, 512, 512, 1],
               ['fwd', 64, 28, 28, 128, 128, 1], ['dgrad', 64, 28, 28, 128, 128, 1], ['fwd', 64, 28, 28, 128, 256, 2],
               ['fwd', 128, 14, 14, 256, 256, 1], ['dgrad', 128, 14, 14, 256, 256, 1], ['fwd', 64, 56, 56, 64, 64, 1]])
    for i in (0, 2, 4, 6, 7):
        _has(cs[i], 'igemm_sk_kernel<64,64,2,2,0,0,0>')
    for i in (1, 3, 5, 8):
        _has(cs[i], 'igemm_sk_kernel<128,64,2,2,0,1,1>')
    assert not any(s.startswith('igemm_sk') for s in cs[9]['symbols']), cs[9]


@pytest.mark.parametrize('tile', ['0', '2', '3'])
def test_stream_k_hooked_every_tile_and_ragged_shapes(tile):
    """FTE_SK=2 puts every eligible launch on stream-K, FTE_SK_TILE picks the tile (0: 128x128, 2: 128x64, 3: 64x64): ragged M (189,
    5291 rows), N = 192 (128x128 falls back to 128x64), fewer iterations than workers (3x9x7: 36-54 one-step workers), K = 576 with
    two K-steps per worker, and the shapes of the natural test on the other tiles."""
    cases = json.load(open(os.path.join(HERE, 'golden', 'sk_cases.json')))
    cs = _run(cases, {'FTE_SK': '2', 'FTE_SK_TILE': tile})
    for c in cs:
        assert any(s.startswith('igemm_sk_kernel<') for s in c['symbols']), c


def test_stream_k_flag_words_in_the_workspace():
    """FTE_SK_WS_FLAGS=1: the fallback form of the hand-over flags (words behind the slabs in the caller's workspace, zeroed by a
    memset in front of the launch) instead of the library's per-stream epoch words."""
    cs = _run([['fwd', 64, 14, 14, 256, 256, 1], ['dgrad', 64, 14, 14, 256, 256, 1]], {'FTE_SK_WS_FLAGS': '1'})
    _has(cs[0], 'igemm_sk_kernel<64,64,2,2,0,0,0>')
    _has(cs[1], 'igemm_sk_kernel<128,64,2,2,0,1,1>')


def test_winograd_symbols_run_naturally_at_the_headline_shapes():
    """FTE_CONV_ALGO=auto (the default): the stride-1 3x3 layers of >= 128 channels take the Winograd kernels for all three products,
    the 64-channel stage for forward and filter gradient (csrc/api.hip wino_sized); each symbol at a shard of the headline shape that
    gives every resident block several tiles and a ragged last row block, block by block against the float64 oracle."""
    cs = _run([['fwd', 72, 14, 14, 256, 256, 1], ['dgrad', 72, 14, 14, 256, 256, 1], ['wgrad', 72, 14, 14, 256, 256, 1],
               ['fwd', 24, 28, 28, 128, 128, 1], ['dgrad', 24, 28, 28, 128, 128, 1], ['wgrad', 24, 28, 28, 128, 128, 1],
               ['fwd', 136, 7, 7, 512, 512, 1], ['dgrad', 136, 7, 7, 512, 512, 1], ['wgrad', 136, 7, 7, 512, 512, 1],
               ['fwd', 8, 56, 56, 64, 64, 1], ['wgrad', 8, 56, 56, 64, 64, 1]], env={'FTE_CONV_ALGO': 'auto'})
    for c in cs:
        want = {'fwd': 'wino_mm_kernel<0,', 'dgrad': 'wino_mm_kernel<1,', 'wgrad': 'wino_wgrad_kernel'}[c['case'][0]]
        assert c['symbols'] and all(s_.startswith(want) for s_ in c['symbols']), c
    # 72 x 14x14 x 256 = 224 tiles: one round of whole tiles; 136 x 7x7 x 512 = 272 tiles: two rounds of whole tiles
    assert cs[0]['symbols'] == ['wino_mm_kernel<0,2>'] and cs[6]['symbols'] == ['wino_mm_kernel<0,2>'], (cs[0], cs[6])
    # the 8-GPU shard of stage 4 (64 images: 128 tiles, half a round): half tiles only, both products
    hs = _run([['fwd', 64, 7, 7, 512, 512, 1], ['dgrad', 64, 7, 7, 512, 512, 1]], env={'FTE_CONV_ALGO': 'auto'})
    assert hs[0]['symbols'] == ['wino_mm_kernel<0,1>'] and hs[1]['symbols'] == ['wino_mm_kernel<1,1>'], hs
    # the A/B hook: the last, partly filled round of a two-round launch as a half-tile launch of its own behind the whole tiles
    ab = _run([['fwd', 136, 7, 7, 512, 512, 1], ['dgrad', 84, 14, 14, 256, 256, 1]], env={'FTE_CONV_ALGO': 'auto', 'FTE_WINO_HALF_TILES': '2'})
    assert sorted(ab[0]['symbols']) == ['wino_mm_kernel<0,1>', 'wino_mm_kernel<0,2>'], ab[0]
    assert sorted(ab[1]['symbols']) == ['wino_mm_kernel<1,1>', 'wino_mm_kernel<1,2>'], ab[1]
    d = _run([['dgrad', 8, 56, 56, 64, 64, 1]], env={'FTE_CONV_ALGO': 'auto'})[0]      # the 64-channel data gradient stays direct
    assert all(s_.startswith('igemm') for s_ in d['symbols']), d


def test_every_conv_symbol_of_the_headline_run_was_checked():
    """tests/golden/headline_symbols.json lists roofline.per_symbol of the committed bench line (profiles/); every conv symbol in
    it must have run -- against the oracle -- in the tests above; the dense symbols (FC / classifier products) are the
    instantiations test_gpu_kernels.py::test_dense_nn_nt_tn runs at (512, 512, 25088) and (70, 10624, 512)."""
    if not SEEN:
        pytest.skip('run the whole module: the earlier tests collect the symbols')
    want = json.load(open(os.path.join(HERE, 'golden', 'headline_symbols.json')))
    missing = [s for s in want['conv_symbols'] if s not in SEEN]
    assert not missing, 'headline conv symbols never compared with the oracle: %s' % missing
