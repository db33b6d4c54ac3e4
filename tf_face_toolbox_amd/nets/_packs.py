"""bf16 packs of a net's conv filters (the operands of the bf16-source convolutions), refreshed once per step with ONE launch per
layout (`fte_pack_weights_bf16_table`) instead of one launch per conv and layout."""
import torch

from .. import _lib


class FilterPacks(object):
    """entries: [(name, arena offset in floats, ksize, cin, cout)].  `w16[name]` is the HWIO pack [k,k,cin,cout] (the data
    gradient's operand), `w16t[name]` the [k,k,cout,cin] pack (forward's)."""

    CHUNK = 64          # rows per table launch (the kernel keeps the table in LDS)

    def __init__(self, entries, device, head=0):
        """`head` > 0: the first `head` entries form launches of their own (refresh_head / refresh_rest): a net whose forward walk
        starts with them can pack those on its main stream and the rest -- and every HWIO pack, which only the backward pass reads --
        on a side stream, under its first layers (nets/graph.py)."""
        i16 = dict(dtype=torch.int16, device=device)
        self.head = int(head)
        off = 0
        rows = []
        for name, src, k, cin, cout in entries:
            size = k * k * cin * cout
            assert cin % 4 == 0 and cout % 4 == 0 and size % 8 == 0, (name, cin, cout)
            rows.append([int(src), off, k * k, cin, cout, 0, name, k])
            off += size
        self.total = off
        self.a16 = torch.empty(max(off, 8), **i16)
        self.a16t = torch.empty(max(off, 8), **i16)
        self.w16, self.w16t = {}, {}
        for src, dst, taps, cin, cout, _, name, k in rows:
            self.w16[name] = self.a16[dst:dst + taps * cin * cout].view(k, k, cin, cout)
            self.w16t[name] = self.a16t[dst:dst + taps * cin * cout].view(k, k, cout, cin)
        self.launches = []          # (device table, rows, elements, destination base offset, all rows inside the head)
        starts = list(range(0, len(rows), self.CHUNK))
        if 0 < self.head < len(rows):
            starts = [0] + list(range(self.head, len(rows), self.CHUNK))
        bounds = starts[1:] + [len(rows)]
        for c0, c1 in zip(starts, bounds):
            chunk = rows[c0:c1]
            base = chunk[0][1]
            start = 0
            tab = []
            for src, dst, taps, cin, cout, _, name, k in chunk:
                tab.append([src, dst - base, taps, cin, cout, start // 4])
                start += taps * cin * cout
            self.launches.append((torch.tensor(tab, dtype=torch.int32, device=device).contiguous(), len(chunk), start, base,
                                  0 < self.head < len(rows) and c1 <= self.head))
        self.head_names = set(r[6] for r in rows[:self.head]) if 0 < self.head < len(rows) else set(r[6] for r in rows)

    def refresh(self, params, st):
        for tab, n, total, base, _ in self.launches:
            _lib.call('fte_pack_weights_bf16_table', params, self.a16[base:], tab, n, total, 0, st)
            _lib.call('fte_pack_weights_bf16_table', params, self.a16t[base:], tab, n, total, 1, st)

    def refresh_head(self, params, st):
        """the forward ([cout][cin]) packs of the head entries"""
        for tab, n, total, base, is_head in self.launches:
            if is_head:
                _lib.call('fte_pack_weights_bf16_table', params, self.a16t[base:], tab, n, total, 1, st)

    def refresh_rest(self, params, st):
        """everything refresh_head left: the forward packs of the other entries, the HWIO packs of all"""
        for tab, n, total, base, is_head in self.launches:
            if not is_head:
                _lib.call('fte_pack_weights_bf16_table', params, self.a16t[base:], tab, n, total, 1, st)
        for tab, n, total, base, _ in self.launches:
            _lib.call('fte_pack_weights_bf16_table', params, self.a16[base:], tab, n, total, 0, st)
