"""Host-side packing of MLP weights into MFMA A-fragment order (and back for gradients).

The fused kernels in csrc/mlp.hip keep activations in the accumulator registers of
v_mfma_f32_32x32x2_f32 between layers.  That fixes, for every layer, (a) which input feature a
k-step/lane-half pair (kk, h) carries and (b) which output row an accumulator element holds:

    accumulator tile t, register r, lane half h  <->  row 32*t + (r&3) + 8*(r>>2) + 4*h
    next layer's k-step kk = 16*t + r            ->  B operand of half h is that same row

so the weight (A) operand of tile mt, k-step kk, lane l must be  W[rowmap[32*mt + (l&31)]][kmap[kk][l>>5]].
Packed layout (all packs):  [mt][q = kk//4][lane 0..63][j = kk%4]  -> one float4 per lane per quad.

All maps are built once in numpy; packing a parameter set is ONE gather over the concatenated
flat weights (a trailing zero is the target of every "no such row/column" entry), and recovering
natural-layout gradients from the weight-gradient kernel's tiles is one gather each for dW and db.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


def acc_row(r: int, h: int) -> int:
    return (r & 3) + 8 * (r >> 2) + 4 * h


def kmap_acc(n_tiles: int) -> np.ndarray:
    """k-step -> feature carried by each lane half when the input is an accumulator set."""
    km = np.zeros((16 * n_tiles, 2), dtype=np.int64)
    for kk in range(16 * n_tiles):
        t, r = kk >> 4, kk & 15
        for h in (0, 1):
            km[kk, h] = 32 * t + acc_row(r, h)
    return km


def kmap_enc20() -> np.ndarray:
    """Frequency-encoding k-steps (csrc/mlp.hip:enc_bin) -> index in the reference's 39-vector
    [x(3), sin(2^0 x)(3), cos(2^0 x)(3), ...] (encodings.py:35-57)."""
    km = -np.ones((20, 2), dtype=np.int64)
    for kk in range(18):
        band, dim = kk // 3, kk % 3
        km[kk, 0] = 3 + band * 6 + dim        # sin
        km[kk, 1] = 3 + band * 6 + 3 + dim    # cos
    km[18] = (0, 1)
    km[19] = (2, -1)
    return km


def rowmapT_kk(kmap: np.ndarray, n_tiles: int) -> np.ndarray:
    """Transposed-layer output rows ordered so that lane half h / register r of tile t holds
    d(feature(kk = 16 t + r, h)): row rho = 32 t + q, h = (q>>2)&1, r = (q&3) + 4 (q>>3)."""
    rm = -np.ones(32 * n_tiles, dtype=np.int64)
    for rho in range(32 * n_tiles):
        t, q = rho >> 5, rho & 31
        h, r = (q >> 2) & 1, (q & 3) + 4 * (q >> 3)
        kk = 16 * t + r
        if kk < kmap.shape[0]:
            rm[rho] = kmap[kk, h]
    return rm


def _frag_index(row_of_lane: np.ndarray, col_of_k: np.ndarray, n_cols: int, transpose: bool, sentinel: int,
                base: int) -> np.ndarray:
    """Gather index [MT][KS/4][64][4] into the flat weight vector."""
    MT, KS = row_of_lane.shape[0] // 32, col_of_k.shape[0]
    idx = np.full((MT, KS // 4, 64, 4), sentinel, dtype=np.int64)
    for mt in range(MT):
        for kk in range(KS):
            q, j = kk // 4, kk % 4
            for lane in range(64):
                a = row_of_lane[mt * 32 + (lane & 31)]
                b = col_of_k[kk, lane >> 5]
                if a < 0 or b < 0:
                    continue
                idx[mt, q, lane, j] = base + ((b * n_cols + a) if transpose else (a * n_cols + b))
    return idx.reshape(-1)


def _frag_index_b3(row_of_lane: np.ndarray, col_of_k: np.ndarray, n_cols: int, transpose: bool, sentinel: int,
                   base: int) -> np.ndarray:
    """Gather index [MT][KS16][64][8] for the bf16x3 kernels (csrc/mlp_b3.hip): with v_mfma_f32_32x32x16_bf16 lane
    (i = lane & 31, g = lane >> 5) supplies A[m = i][k = 8g + e] of k16-step s, and the B operand of (s, g, e) is what the
    fp32 chain carries in k-step 8s + e, lane half g -- so the same kmap serves both fragment shapes."""
    MT, KS = row_of_lane.shape[0] // 32, col_of_k.shape[0]
    KS16 = (KS + 7) // 8
    km = np.full((KS16 * 8, 2), -1, dtype=np.int64)
    km[:KS] = col_of_k
    lane = np.arange(64)
    a = row_of_lane.reshape(MT, 32)[:, lane & 31]                        # [MT, 64]
    b = km.reshape(KS16, 8, 2)[:, :, lane >> 5].transpose(0, 2, 1)       # [KS16, 64, 8]
    a4, b4 = a[:, None, :, None], b[None, :, :, :]
    flat = (b4 * n_cols + a4) if transpose else (a4 * n_cols + b4)
    idx = np.where((a4 >= 0) & (b4 >= 0), base + flat, sentinel)
    return idx.reshape(-1)


def _b3_blocks(idx: np.ndarray, MT: int, KS16: int):
    """Split a layer's b3 fragment index into the blocks the kernels stage at a time.  A 128 x 128 layer (MT = 4, KS16 = 8)
    is staged as two k-halves [khalf][out tile][k16 step 0..3][lane][8] of 48 KB of slices each (the phased kernels of
    csrc/mlp_b3.hip keep three such blocks in LDS); every other layer is one block."""
    if MT == 4 and KS16 == 8:
        v = idx.reshape(4, 2, 4, 64, 8).transpose(1, 0, 2, 3, 4)
        return [v[0].reshape(-1), v[1].reshape(-1)]
    return [idx]


B3_DMA_F4 = 512      # a layer's slices are staged by whole rounds of the 512-thread block (16 bytes per thread and round)


@dataclass
class LayerSpec:
    """One linear layer as the kernels see it."""
    name: str
    out_dim: int                  # natural rows of W
    in_dim: int                   # natural columns of W
    kmap: np.ndarray              # [KS,2] forward: k-step -> input column (or -1)
    rowmap: np.ndarray            # [32*MT] forward: accumulator row -> output row (or -1)
    kmapT: np.ndarray             # [KS',2] backward: k-step -> output row (or -1)
    rowmapT: np.ndarray           # [32*MT'] backward: accumulator row -> input column (or -1)
    act_rows: np.ndarray          # [in_pad] weight-grad: act tile row -> input column (or -1)
    dpre_rows: np.ndarray         # [out_pad] weight-grad: dpre tile row -> output row (or -1)


def _ident(n: int, pad: int) -> np.ndarray:
    v = -np.ones(pad, dtype=np.int64)
    v[:n] = np.arange(n)
    return v


def _kk_rows(kmap: np.ndarray, pad: int) -> np.ndarray:
    v = -np.ones(pad, dtype=np.int64)
    for kk in range(kmap.shape[0]):
        for h in (0, 1):
            v[2 * kk + h] = kmap[kk, h]
    return v


def warp_net_specs(n_out: int) -> List[LayerSpec]:
    """deform_net (n_out=3) / topo_net (n_out=2): 39(+code as bias) -> 128 x5 -> n_out."""
    enc = kmap_enc20()
    a128 = kmap_acc(4)
    a32 = kmap_acc(1)
    specs = [LayerSpec("l0", 128, 39, enc, _ident(128, 128), a128, rowmapT_kk(enc, 2), _kk_rows(enc, 64),
                       _ident(128, 128))]
    for l in range(1, 5):
        specs.append(LayerSpec(f"l{l}", 128, 128, a128, _ident(128, 128), a128, _ident(128, 128), _ident(128, 128),
                               _ident(128, 128)))
    kT = np.where(a32 < n_out, a32, -1)
    specs.append(LayerSpec("l5", n_out, 128, a128, _ident(n_out, 32), kT, _ident(128, 128), _ident(128, 128),
                           _ident(n_out, 32)))
    return specs


def field_specs() -> List[LayerSpec]:
    """sdf_net 73->64->64->33 and color_net 64->64->64->3 (model.py:273-307)."""
    enc = kmap_enc20()
    a64 = kmap_acc(2)
    a32 = kmap_acc(1)
    # sdf L0 input columns: [enc 0..38 | hash 39..70 | topo 71..72]
    k0 = -np.ones((40, 2), dtype=np.int64)
    k0[:20] = enc
    for kk in range(20, 36):
        k0[kk] = (39 + (kk - 20), 39 + 16 + (kk - 20))
    k0[36] = (71, 72)
    rT0 = -np.ones(96, dtype=np.int64)
    rT0[:64] = rowmapT_kk(enc, 2)
    for h in (0, 1):                                   # tile 1, register 4: topo[h]
        rT0[32 + acc_row(4, h)] = 71 + h
    for q in range(32):                                # tile 2: hash feature 16 h + r
        h, r = (q >> 2) & 1, (q & 3) + 4 * (q >> 3)
        rT0[64 + q] = 39 + 16 * h + r
    s0 = LayerSpec("s0", 64, 73, k0, _ident(64, 64), a64, rT0, _kk_rows(k0, 96), _ident(64, 64))
    s1 = LayerSpec("s1", 64, 64, a64, _ident(64, 64), a64, _ident(64, 64), _ident(64, 64), _ident(64, 64))
    # sdf L2 rows: tile 0 = geo (rows 1..32), tile 1 row 0 = sdf
    r2 = -np.ones(64, dtype=np.int64)
    r2[:32] = 1 + np.arange(32)
    r2[32] = 0
    k2T = -np.ones((32, 2), dtype=np.int64)
    for kk in range(16):
        for h in (0, 1):
            k2T[kk, h] = 1 + acc_row(kk, h)
            k2T[16 + kk, h] = 0 if acc_row(kk, h) == 0 else -1
    s2 = LayerSpec("s2", 33, 64, a64, r2, k2T, _ident(64, 64), _ident(64, 64), r2.copy())
    # color L0 input columns: [hash_c 0..31 | geo 32..63]
    kc = -np.ones((32, 2), dtype=np.int64)
    for kk in range(16):
        kc[kk] = (kk, 16 + kk)
        for h in (0, 1):
            kc[16 + kk, h] = 32 + acc_row(kk, h)
    rTc = -np.ones(64, dtype=np.int64)
    for q in range(32):
        h, r = (q >> 2) & 1, (q & 3) + 4 * (q >> 3)
        rTc[q] = 16 * h + r                            # tile 0: hash_c feature 16 h + r
        rTc[32 + q] = 32 + q                           # tile 1: geo in accumulator order
    c0 = LayerSpec("c0", 64, 64, kc, _ident(64, 64), a64, rTc, _kk_rows(kc, 64), _ident(64, 64))
    c1 = LayerSpec("c1", 64, 64, a64, _ident(64, 64), a64, _ident(64, 64), _ident(64, 64), _ident(64, 64))
    kT = np.where(a32 < 3, a32, -1)
    c2 = LayerSpec("c2", 3, 64, a64, _ident(3, 32), kT, _ident(64, 64), _ident(64, 64), _ident(3, 32))
    return [s0, s1, s2, c0, c1, c2]


class NetPacker:
    """Index maps for a list of layers sharing one flat weight vector [W_0 | W_1 | ... | 0]."""

    def __init__(self, specs: Sequence[LayerSpec], bwd_order: Sequence[int]):
        self.specs = list(specs)
        offs, n = [], 0
        for s in self.specs:
            offs.append(n)
            n += s.out_dim * s.in_dim
        self.w_offsets, self.n_weights = offs, n
        sentinel = n
        fwd = [_frag_index(s.rowmap, s.kmap, s.in_dim, False, sentinel, o) for s, o in zip(self.specs, offs)]
        bwd = [_frag_index(self.specs[i].rowmapT, self.specs[i].kmapT, self.specs[i].in_dim, True, sentinel, offs[i])
               for i in bwd_order]
        self.fwd_index = np.concatenate(fwd)
        self.bwd_index = np.concatenate(bwd)
        # bf16x3 fragments: fp32 gather in b3 order (sliced into three bf16 planes by mh_b3_slice); per layer the float count
        # and the float4 offset of its [hi|mid|lo] planes in the sliced pack (padded to whole DMA rounds)
        fwd3, self.fwd3_layer = [], []          # fwd3_layer: which layer of the chain a staged block belongs to
        for li, (s, o) in enumerate(zip(self.specs, offs)):
            blocks = _b3_blocks(_frag_index_b3(s.rowmap, s.kmap, s.in_dim, False, sentinel, o), s.rowmap.shape[0] // 32,
                                (s.kmap.shape[0] + 7) // 8)
            fwd3 += blocks
            self.fwd3_layer += [li] * len(blocks)
        self.fwd3_index = np.concatenate(fwd3)
        self.fwd3_n = [len(f) for f in fwd3]
        self.fwd3_f4, f4 = [], 0
        for n in self.fwd3_n:
            self.fwd3_f4.append(f4)
            f4 += -(-(3 * n // 8) // B3_DMA_F4) * B3_DMA_F4
        self.fwd3_total_f4 = f4
        bwd3, self.bwd3_layer = [], []
        for li, i in enumerate(bwd_order):
            sp = self.specs[i]
            blocks = _b3_blocks(_frag_index_b3(sp.rowmapT, sp.kmapT, sp.in_dim, True, sentinel, offs[i]), sp.rowmapT.shape[0] // 32,
                                (sp.kmapT.shape[0] + 7) // 8)
            bwd3 += blocks
            self.bwd3_layer += [li] * len(blocks)
        self.bwd3_index = np.concatenate(bwd3)
        self.bwd3_n = [len(f) for f in bwd3]
        self.bwd3_f4, f4 = [], 0
        for n in self.bwd3_n:
            self.bwd3_f4.append(f4)
            f4 += -(-(3 * n // 8) // B3_DMA_F4) * B3_DMA_F4
        self.bwd3_total_f4 = f4
        # bias vector (accumulator-row order, padded to 32*MT) : gather from [b_0 | b_1 | ... | 0]
        b_offs, nb = [], 0
        for s in self.specs:
            b_offs.append(nb)
            nb += s.out_dim
        self.b_offsets, self.n_biases = b_offs, nb
        bidx = []
        for s, o in zip(self.specs, b_offs):
            bidx.append(np.where(s.rowmap >= 0, s.rowmap + o, nb))
        self.bias_index = bidx
        # weight-grad tiles -> natural layout
        dw_idx, db_idx, raw_off, rawb_off = [], [], 0, 0
        self.wg_in, self.wg_out = [], []
        for s in self.specs:
            in_pad, out_pad = s.act_rows.shape[0], s.dpre_rows.shape[0]
            arow = -np.ones(s.in_dim, dtype=np.int64)
            for i, c in enumerate(s.act_rows):
                if c >= 0:
                    arow[c] = i
            drow = -np.ones(s.out_dim, dtype=np.int64)
            for i, r in enumerate(s.dpre_rows):
                if r >= 0:
                    drow[r] = i
            assert (arow >= 0).all() and (drow >= 0).all(), s.name
            dw_idx.append((raw_off + drow[:, None] * in_pad + arow[None, :]).reshape(-1))
            db_idx.append(rawb_off + drow)
            raw_off += in_pad * out_pad
            rawb_off += out_pad
            self.wg_in.append(in_pad)
            self.wg_out.append(out_pad)
        self.dw_index = np.concatenate(dw_idx)
        self.db_index = np.concatenate(db_idx)
        self.raw_dw, self.raw_db = raw_off, rawb_off
        self._dev: Dict[Tuple[str, int], Dict[str, torch.Tensor]] = {}

    def on(self, device: torch.device) -> Dict[str, torch.Tensor]:
        key = (device.type, device.index or 0)
        if key not in self._dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev[key] = dict(fwd=t(self.fwd_index), bwd=t(self.bwd_index), dw=t(self.dw_index), db=t(self.db_index),
                                  bias=[t(b) for b in self.bias_index])
        return self._dev[key]

    # ---- tensors
    def flat_weights(self, weights: Sequence[torch.Tensor]) -> torch.Tensor:
        parts = [w.reshape(-1) for w in weights]
        return torch.cat(parts + [parts[0].new_zeros(1)])

    def pack(self, weights: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        m = self.on(weights[0].device)
        flat = self.flat_weights(weights)
        return flat[m["fwd"]].contiguous(), flat[m["bwd"]].contiguous()

    def pack_biases(self, biases: Sequence[torch.Tensor], skip_first: bool) -> torch.Tensor:
        m = self.on(biases[0].device)
        flat = torch.cat([b.reshape(-1) for b in biases] + [biases[0].new_zeros(1)])
        sel = m["bias"][1:] if skip_first else m["bias"]
        return torch.cat([flat[i] for i in sel]).contiguous()

    def unpack_grads(self, dw_raw: torch.Tensor, db_raw: torch.Tensor):
        """dw_raw [raw_dw], db_raw [raw_db] (already reduced over chunks) -> lists of natural dW, db."""
        m = self.on(dw_raw.device)
        dw = dw_raw[m["dw"]]
        db = db_raw[m["db"]]
        outw, outb, o, ob = [], [], 0, 0
        for s in self.specs:
            n = s.out_dim * s.in_dim
            outw.append(dw[o:o + n].view(s.out_dim, s.in_dim))
            outb.append(db[ob:ob + s.out_dim])
            o += n
            ob += s.out_dim
        return outw, outb


class JointPacker:
    """Several NetPackers behind ONE gather each way, so that preparing the kernels' operands costs 4 launches per call
    instead of ~15 per net (the maps are static; only the number of launches changes, not a single value):

      flat in   : [W of net 0 | W of net 1 | ... | b of net 0 | b of net 1 | ... | 0]        (natural, effective)
      fwd pack  : [A-fragments net 0 | net 1 | ... | bias pack net 0 | net 1 | ...]          -> `w[k]`, `b[k]` slices
      bwd pack  : [transposed fragments net 0 | net 1 | ...]                                 -> `wT[k]` slices
      raw grads : [dW tiles net 0 | net 1 | ... | db tiles net 0 | net 1 | ...] (mh_mlp_wgrad's output) ->
      natural   : [dW net 0 | dW net 1 | ... | db net 0 | db net 1 | ...]
    """

    def sliced_bwd_for(self, mode) -> bool:
        """are the transposed layers sliced in arithmetic mode `mode` ("b3"; the kernels' b3 flag spells it as True)?"""
        mode = "b3" if mode is True else mode
        return self.sliced_bwd and mode == "b3"

    def __init__(self, packers: Sequence[NetPacker], skip_first_bias: bool, sliced_bwd: bool = True):
        self.packers = list(packers)
        # sliced_bwd: do the sliced (bf16 x 3) kernels read TRANSPOSED slices of these nets?  (The f32 mode reads the fp32
        # transposed pack only: sliced_bwd_for)
        self.sliced_bwd = sliced_bwd
        nW = sum(p.n_weights for p in self.packers)
        nB = sum(p.n_biases for p in self.packers)
        zero = nW + nB
        fwd, bwd, bias = [], [], []
        self.w, self.wT, self.b = [], [], []
        wo, bo, f_off, t_off = 0, nW, 0, 0
        for p in self.packers:
            fwd.append(np.where(p.fwd_index == p.n_weights, zero, p.fwd_index + wo))
            bwd.append(np.where(p.bwd_index == p.n_weights, zero, p.bwd_index + wo))
            sel = p.bias_index[1:] if skip_first_bias else p.bias_index
            bi = np.concatenate(sel)
            bias.append(np.where(bi == p.n_biases, zero, bi + bo))
            self.w.append((f_off, len(p.fwd_index)))
            self.wT.append((t_off, len(p.bwd_index)))
            f_off += len(p.fwd_index)
            t_off += len(p.bwd_index)
            wo += p.n_weights
            bo += p.n_biases
        for bi in bias:
            self.b.append((f_off, len(bi)))
            f_off += len(bi)
        assert all(o % 4 == 0 for o, _ in self.w + self.wT + self.b), "kernel operands are read as float4"
        self.fwd_index = np.concatenate(fwd + bias)
        self.bwd_index = np.concatenate(bwd)
        self.n_flat = zero + 1
        # bf16x3 forward fragments of all nets: one fp32 gather + one slicing launch (mh_b3_slice)
        f3, self.w3, self.b3_layers, wo, src, f4 = [], [], [], 0, 0, 0
        for p in self.packers:
            f3.append(np.where(p.fwd3_index == p.n_weights, zero, p.fwd3_index + wo))
            self.w3.append((f4, p.fwd3_total_f4))                       # float4 units
            for n, o4 in zip(p.fwd3_n, p.fwd3_f4):
                self.b3_layers.append((src, n, f4 + o4))
                src += n
            f4 += p.fwd3_total_f4
            wo += p.n_weights
        self.fwd3_index = np.concatenate(f3)
        self.fwd3_total_f4 = f4
        t3, self.wT3, self.b3T_layers, wo, src, f4 = [], [], [], 0, 0, 0
        for p in self.packers:
            t3.append(np.where(p.bwd3_index == p.n_weights, zero, p.bwd3_index + wo))
            self.wT3.append((f4, p.bwd3_total_f4))
            for n, o4 in zip(p.bwd3_n, p.bwd3_f4):
                self.b3T_layers.append((src, n, f4 + o4))
                src += n
            f4 += p.bwd3_total_f4
            wo += p.n_weights
        self.bwd3_index = np.concatenate(t3)
        self.bwd3_total_f4 = f4
        # gradients
        raw_dw = sum(p.raw_dw for p in self.packers)
        g, dwo, dbo = [], 0, raw_dw
        for p in self.packers:
            g.append(p.dw_index + dwo)
            dwo += p.raw_dw
        for p in self.packers:
            g.append(p.db_index + dbo)
            dbo += p.raw_db
        self.grad_index = np.concatenate(g)
        self.raw_len = dbo
        # raw offsets of the FIRST layer's bias gradient of every net (tile-row order == natural order for these layers),
        # and a copy of the gradient map in which those entries read a zero sentinel appended to the raw vector: the warp
        # nets' first-layer bias reaches its parameter through the per-frame bias0, not through the raw gradient
        self.bias0_raw, off = [], raw_dw
        for p in self.packers:
            self.bias0_raw.append(off)
            off += p.raw_db
        gz = self.grad_index.copy()
        pos = sum(p.n_weights for p in self.packers)
        for p in self.packers:
            gz[pos:pos + p.specs[0].out_dim] = self.raw_len
            pos += p.n_biases
        self.grad_index_nob0 = gz
        self._dev: Dict[Tuple[str, int], Dict[str, torch.Tensor]] = {}

    def on(self, device: torch.device) -> Dict[str, torch.Tensor]:
        key = (device.type, device.index or 0)
        if key not in self._dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev[key] = dict(fwd=t(self.fwd_index), bwd=t(self.bwd_index), grad=t(self.grad_index),
                                  grad_nob0=t(self.grad_index_nob0), fwd3=t(self.fwd3_index), bwd3=t(self.bwd3_index),
                                  all=t(self.all_index()[0]))
        return self._dev[key]

    def all_index(self):
        """-> (index, {section: (offset, length)}): the four operand gathers (fwd | bwd | fwd3 | bwd3) as ONE gather of the flat
        parameter vector, every section padded with the zero sentinel to a whole float4 (the kernels read their operands as
        float4, mh_b3_slice takes 16-byte aligned sources)"""
        if getattr(self, "_all", None) is None:
            zero = self.n_flat - 1
            parts, sect, off = [], {}, 0
            for name, idx in (("fwd", self.fwd_index), ("bwd", self.bwd_index), ("fwd3", self.fwd3_index), ("bwd3", self.bwd3_index)):
                pad = (-len(idx)) % 4
                parts.append(np.concatenate([idx, np.full(pad, zero, dtype=idx.dtype)]))
                sect[name] = (off, len(idx))
                off += len(idx) + pad
            self._all = (np.concatenate(parts), sect)
        return self._all

    def pack(self, weights: Sequence[Sequence[torch.Tensor]], biases: Sequence[Sequence[torch.Tensor]]):
        """weights[k] / biases[k]: the layers of net k (natural).  -> (fwd pack, bwd pack); slice with self.w/b/wT."""
        parts = [w.reshape(-1) for net in weights for w in net] + [b.reshape(-1) for net in biases for b in net]
        flat = torch.cat(parts + [parts[0].new_zeros(1)])
        assert flat.numel() == self.n_flat
        m = self.on(flat.device)
        return flat[m["fwd"]], flat[m["bwd"]]

    def flat(self, weights, biases) -> torch.Tensor:
        parts = [w.reshape(-1) for net in weights for w in net] + [b.reshape(-1) for net in biases for b in net]
        flat = torch.cat(parts + [parts[0].new_zeros(1)])
        assert flat.numel() == self.n_flat
        return flat

    @staticmethod
    def take(buf: torch.Tensor, sl: Tuple[int, int]) -> torch.Tensor:
        return buf[sl[0]:sl[0] + sl[1]]

    def unpack_grads(self, raw: torch.Tensor, zero_bias0: bool = False):
        """raw [raw_len] = mh_mlp_wgrad's dw_raw | db_raw -> per net (list of natural dW, list of natural db), views of
        one gathered buffer.  zero_bias0: the first layer's bias gradient of every net comes out as zeros."""
        assert raw.numel() == self.raw_len
        if zero_bias0:
            nat = torch.cat([raw, raw.new_zeros(1)])[self.on(raw.device)["grad_nob0"]]
        else:
            nat = raw[self.on(raw.device)["grad"]]
        out_w, out_b, o = [], [], 0
        for p in self.packers:
            ws = []
            for s in p.specs:
                n = s.out_dim * s.in_dim
                ws.append(nat[o:o + n].view(s.out_dim, s.in_dim))
                o += n
            out_w.append(ws)
        for p in self.packers:
            bs = []
            for s in p.specs:
                bs.append(nat[o:o + s.out_dim])
                o += s.out_dim
            out_b.append(bs)
        return out_w, out_b


_PACKERS: Dict[str, object] = {}


def warp_packer(n_out: int) -> NetPacker:
    k = f"warp{n_out}"
    if k not in _PACKERS:
        # transposed packs in the order the backward kernel consumes them: T5, T4, T3, T2, T1, T0
        _PACKERS[k] = NetPacker(warp_net_specs(n_out), bwd_order=[5, 4, 3, 2, 1, 0])
    return _PACKERS[k]


def field_packer() -> NetPacker:
    if "field" not in _PACKERS:
        # backward consumption order: TC2, TC1, TC0, TS2, TS1, TS0
        _PACKERS["field"] = NetPacker(field_specs(), bwd_order=[5, 4, 3, 2, 1, 0])
    return _PACKERS["field"]


def warp_joint_packer() -> JointPacker:
    if "warp_joint" not in _PACKERS:
        # b0 of both nets lives in the per-slot bias0 the caller builds (ops._WarpMLP), not in the bias pack
        _PACKERS["warp_joint"] = JointPacker([warp_packer(3), warp_packer(2)], skip_first_bias=True)
    return _PACKERS["warp_joint"]


def field_joint_packer() -> JointPacker:
    if "field_joint" not in _PACKERS:
        _PACKERS["field_joint"] = JointPacker([field_packer()], skip_first_bias=False)
    return _PACKERS["field_joint"]
