"""profiles/<tag>_hbm_bound_kernels.md from the round's rocprofv3 evidence, at the shard sizes the BASELINE configs run:

  * the BN nets (scripts/collect_net_profiles.sh): profiles/<tag>_<net>_pmc_summary.csv -- per kernel symbol the average duration
    of the `--kernel-trace` pass and the HBM bytes per launch of the FETCH_SIZE / WRITE_SIZE passes (2 x FETCH + WRITE, KiB:
    /opt/skills/guides/MI355X_MICROARCH.md's gfx950 correction), programs placed directly after `--`;
  * SphereNet in the three precision modes (scripts/collect_profiles.sh): profiles/<tag>[_mode]_pmc_hbm_traffic_summary.csv + the
    kernel stats of the same command.

Every symbol that the PMC summary classes as HBM-bound (its MFMA pipe is idle or absent) and that takes >= 1 % of the config's
kernel time is listed with its achieved share of 8 TB/s; below 55 % the reason column says why (REASONS below: measured causes,
each with the experiment that established it in DESIGN.md / profiles/<tag>_notes.md).

    python scripts/hbm_table.py r4 > profiles/r4_hbm_bound_kernels.md"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8000.0

REASONS = [
    (r'bn_bwd_reduce_v4', 'two tensors read, nothing written: 25-50 MB per launch = 10-15 us, of which ~5 us are launch + first-byte latency; '
                          '4 rows in flight per lane (8 KB per block and tensor) is what the register budget allows at 8 blocks per CU'),
    (r'bn_bwd_apply|bn_apply', 'streams at 4.5-5.5 TB/s on the 56x56 / 28x28 layers; the 7x7 / 4x4 layers move 8-25 MB in 7-9 us -- launch-latency floor'),
    (r'bn_(bwd_)?finalize', 'not a streaming kernel: sums <= 2048 partial rows per channel (0.1-1.5 MB), 5-6 us = launch latency'),
    (r'reduce_slabs|reduce_rows|gconv_wgrad16_reduce', 'split-K / partial-row reductions of 0.5-20 MB: 5-15 us each, latency-bound (side stream: overlapped)'),
    (r'wgrad16p|wgrad16_kernel', 'classed HBM-bound by its idle MFMA pipe: a K = pixels product split over <= 768 blocks; its bytes are re-reads '
                                'of the [pixels, cin] / [pixels, cout] operands per output tile (L2 hits, counted by FETCH_SIZE only when they miss)'),
    (r'gconv3x3_wgrad', '32-channel slices through the transposing LDS read; VALU-bound on address and conversion work (side stream: overlapped)'),
    (r'gconv3x3_mfma16', 'VALU-bound, not byte-bound: ~600 VALU instructions per 18 MFMAs (edge masks of 28-wide images inside 32-pixel tiles, '
                         'bf16 <-> fp32 of the BN fold, 2-byte stores); two waves per SIMD at 208-228 registers'),
    (r'pw16_kernel', 'instruction-issue bound beside its memory waits, not byte-bound: SQ counters of the kernel alone (profiles/r5_pw16_sq_counters.csv) -- of a wave\'s '
                     'cycles 37-41 % issue instructions (0.3-0.7 G VALU per 55 launches: bf16 <-> fp32 of the loader transform and the statistics, the '
                     '2-byte writes into the LDS result stage), 38-43 % are parked on s_waitcnt / barriers, 18-24 % are issue stalls (LDS: 5-10 %); at the '
                     'two waves per SIMD the LDS allows (filter slice + a stage per wave) nothing covers the parked share'),
    (r'se_squeeze|se_bwd_gate', 'per-image column sums (one block per image and 64 channels): the 28x28 / 14x14 tensors stream at 3.6-4.2 TB/s with four rows in flight '
                                'per lane; the 7x7 / 4x4 ones are 10-50 MB in 7-12 us -- launch-latency floor'),
    (r'se_apply|se_bn_apply', 'streams at 4.4-5.7 TB/s on the 28x28 / 14x14 tensors; 7x7 / 4x4: 26-40 MB in 9-11 us -- launch-latency floor'),
    (r'se_bn_coef|dense_small', 'not streaming kernels: [images, channels] vectors and the gate\'s 0.1-8 MB dense layers, 6-10 us = launch + one dependent chain'),
    (r'channel_gather_lds', 'whole rows in as 16-byte pieces, the permutation applied by LDS reads, 16-byte stores (round 5; the element gather ran '
                            'at 37 %): 20-100 MB per launch, two barriers per group of rows'),
    (r'channel_gather', 'gathers 4-byte elements at a channel permutation: a 128-byte line serves 32 lanes of one pixel only when the permutation '
                        'keeps neighbours together (ShuffleNet\'s shuffle does not)'),
    (r'dwconv3x3', 'nine taps of a 4-byte element per output, served from L1 / L2; the 14x14 / 7x7 layers are 10-25 MB = launch-latency floor'),
    (r'momentum|adam', 'five streams (w, acc, g read; w, acc written) of the whole arena: runs at 3.7-3.9 TB/s, bounded by the mixed read / write turnaround'),
    (r'pack_weights|gconv_pack16', 'side stream, once per step: transposes every filter into the bf16 table (strided reads)'),
    (r'im2col_first', '7x7x3 patches: 147 of 160 columns gathered from 12-byte pixels, 2-byte stores'),
    (r'maxpool', 'nine overlapping taps per output (L1 hits); one launch per step'),
    (r'conv_first', '3-channel stem on the vector ALUs / 32x32x2 MFMA: bound by its own gather (12-byte pixels), not by HBM'),
    (r'softmax|asoftmax|center|triplet|sum_kernel|gap_|dropout|add_scaled|scale_mask|norms', 'loss-head kernels over [shard, classes] or [shard, 512]: < 100 MB, latency-bound'),
    (r'igemm|pgemm', 'MFMA kernel whose pipe-busy share is under the HBM share for this shape (short K): see the step roofline of the config'),
]


def reason(name):
    for pat, txt in REASONS:
        if re.search(pat, name):
            return txt
    return ''


def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', n).strip('" ')


def net_table(path, title):
    rows = list(csv.DictReader(open(path)))
    out = ['### %s' % title, '', '`%s`' % os.path.relpath(path, ROOT), '',
           '| kernel | launches (run) | avg us | share of kernel time | HBM MB / launch (PMC) | GB/s | of 8 TB/s | why below 55 % |', '|---|---|---|---|---|---|---|---|']
    key = [k for k in rows[0] if k.startswith('hbm_MB')][0]
    n = 0
    for r in rows:
        if r.get('bound', 'hbm') != 'hbm' or float(r['share_of_kernel_time']) < 0.01:
            continue
        if '__amd_rocclr_' in r['kernel'] or 'at::native::' in r['kernel']:      # set-up only (parameter initialisation, host-to-device copies of the
            continue                                                              # initial values): absent from the per-step summaries (*_step_summary.txt)
        frac = float(r['frac_of_8TBps'])
        out.append('| `%s` | %s | %.1f | %.1f %% | %.1f | %.0f | %.0f %% | %s |' % (
            short(r['kernel']), r['dispatches'], float(r['avg_us']), 100 * float(r['share_of_kernel_time']), float(r[key]),
            float(r['achieved_GBps']), 100 * frac, reason(r['kernel']) if frac < 0.55 else ''))
        n += 1
    out.append('')
    return out, n


def sphere_table(pmc, stats, title):
    dur = {}
    tot = 0.0
    for r in csv.DictReader(open(stats)):
        dur[short(r['Name'])] = (float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']))
        tot += float(r['TotalDurationNs'])
    out = ['### %s' % title, '', '`%s` + `%s`' % (os.path.relpath(pmc, ROOT), os.path.relpath(stats, ROOT)), '',
           '| kernel | avg us | share of kernel time | HBM MB / launch (PMC) | algorithmic MB | GB/s | of 8 TB/s | MFMA pipe busy | why below 55 % |',
           '|---|---|---|---|---|---|---|---|---|']
    n = 0
    for r in csv.DictReader(open(pmc)):
        k = short(r['kernel'])
        if k not in dur:
            continue
        us, total = dur[k]
        share = total / tot
        busy = float(r['mfma_pipe_busy_frac'] or 0)
        mb = float(r[[c for c in r if c.startswith('hbm_MB')][0]])
        gbs = mb / us * 1e3
        if share < 0.01 or busy >= 0.25:           # the MFMA-bound symbols have their own roofline (bench.py's `roofline`)
            continue
        out.append('| `%s` | %.1f | %.1f %% | %.1f | %s | %.0f | %.0f %% | %.2f | %s |' % (
            k, us, 100 * share, mb, r['algorithmic_MB_per_launch'], gbs, 100 * gbs / PEAK, busy, reason(k) if gbs / PEAK < 0.55 else ''))
        n += 1
    if not n:
        out.append('| (none: every symbol with >= 1 % of the kernel time keeps its MFMA pipe busy >= 25 % of its duration -- MFMA-bound, `bench.py`\'s `roofline`) | | | | | | | | |')
    out.append('')
    return out, n


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r4'
    prof = os.path.join(ROOT, 'profiles')
    out = ['# HBM-bound kernels at the shard sizes of the BASELINE configs (%s)' % tag, '',
           'Generated by `scripts/hbm_table.py %s` from the rocprofv3 passes named in each section (kernel durations: `--kernel-trace`; bytes: the' % tag,
           '`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command, 2 x FETCH + WRITE KiB per launch -- the gfx950 correction of',
           '`/opt/skills/guides/MI355X_MICROARCH.md`).  Listed: every symbol with >= 1 % of the config\'s kernel time whose bound is HBM (its MFMA',
           'pipe is idle or absent; SphereNet: busy < 25 % of the duration, from the SQ_BUSY_CYCLES / SQ_INSTS_VALU_MFMA pass; its durations come from',
           '`bench.py` under the profiler, whose per-launch event records lengthen the 4-10 us reductions).  "of 8 TB/s" = PMC bytes per launch / average duration / 8000 GB/s.  A launch of 8-50 MB lasts 7-15 us, of which',
           '~5 us are launch and first-byte latency: the small-shard configs (128 images per GPU) sit on that floor, which is what the reason column',
           'says wherever a symbol is below 55 %.  ATen fill kernels and `__amd_rocclr_copyBuffer` are left out: they run while the net is built',
           '(parameter initialisation), never inside a step (`profiles/%s_*_step_summary.txt` list every kernel of a step).' % tag, '']
    nets = [('%s_resnext50_bf16s_b128' % tag, 'config 3: ResNeXt-50 + center loss, bf16 storage, 128 images per GPU'),
            ('%s_senet50_bf16s_b128' % tag, 'config 4: SE-ResNet-50 + triplet loss, bf16 storage, 128 images per GPU'),
            ('%s_shufflenet_b256' % tag, 'config 5: ShuffleNet-v2 x2, fp32, 256 images per GPU'),
            ('%s_resnet50_bf16s_b128' % tag, 'ResNet-50 (the family\'s plain member), bf16 storage, 128 images per GPU')]
    for name, title in nets:
        p = os.path.join(prof, name + '_pmc_summary.csv')
        if os.path.exists(p):
            t, n = net_table(p, title)
            out += t
    for mode, title in (('', 'configs 1-2: SphereNet-20 + A-Softmax, fp32, 512 images per GPU (the headline)'),
                        ('_bf16', 'SphereNet-20, bf16 operands / fp32 tensors, 512 images'),
                        ('_bf16s', 'SphereNet-20, bf16 storage, 512 images')):
        pmc = os.path.join(prof, '%s%s_pmc_hbm_traffic_summary.csv' % (tag, mode))
        st = os.path.join(prof, '%s%s_bench_kernel_stats.csv' % (tag, mode))
        if os.path.exists(pmc) and os.path.exists(st):
            t, n = sphere_table(pmc, st, title)
            out += t
    print('\n'.join(out))


if __name__ == '__main__':
    main()
