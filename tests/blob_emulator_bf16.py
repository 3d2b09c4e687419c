"""Lane-level numpy emulation of mlp_bf16.hip's dataflow, driven by a packed bf16 blob.

Checks the host packer (pack_bf16 in mlp_bf16.hip) and the kernel's operand bookkeeping without a GPU: it walks the weight
stream quad by quad in the kernel's order (output-tile-major jobs of 16 features, the density / colour tiles, the padded tail),
applies v_mfma_f32_16x16x32_bf16 semantics
    A fragment: lane l (i = l&15, q = l>>4), element j -> A[i][k = 8q + j]
    B fragment: lane l, element j                      -> B[k = 8q + j][col = l&15]
    D:          lane l, register r                     -> D[row = 4(l>>4) + r][col = l&15]
and the same packing (the four accumulator registers of tiles 2s, 2s+1 -> elements 0..3 / 4..7 of fragment s, rounded to bf16,
optional ReLU).
Accumulation is fp64: this is a layout check with the kernel's ROUNDING POINTS (bf16 weights, bf16 activations, bf16 encoded
inputs), not a model of fp32 summation order.
"""
import numpy as np

QUAD_ELEMS = 512        # bf16 elements per 1 KiB quad
LANE = np.arange(64)
COL, Q4 = LANE & 15, LANE >> 4
MT, KF = 16, 32


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, elementwise."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


class EmuBf16:
    def __init__(self, blob: np.ndarray):
        h = np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)
        assert h[0] == 0x4D494E46 and h[12] == 2 and h[1] == 3
        self.D, self.W, self.skip, self.L_x, self.L_d = int(h[2]), int(h[3]), int(np.int32(h[4])), int(h[5]), int(h[6])
        so, sb, sdo, sf = int(h[7]), int(h[8]), int(h[10]), int(h[11])
        self.stream = _bf16_to_f32(np.frombuffer(blob[so:so + sb].tobytes(), dtype=np.uint16)).astype(np.float64)
        self.n_quads = sb // 1024
        self.side = np.frombuffer(blob[sdo:sdo + 4 * sf].tobytes(), dtype=np.float32).astype(np.float64)
        D, W, in_d = self.D, self.W, 3 + 6 * self.L_d
        o, f = {}, 0
        for name, n in (("bias_trunk", D * W), ("bias_feat", W), ("bias_d", W // 2), ("head_b", 4), ("wdir_t", in_d * (W // 2))):
            o[name] = f
            f += n
        self.off = o
        self.pos = 0

    # one job: output tile (16 rows) over the k-steps of `frags` ([KS][64 lanes][8]); cin [4][64] accumulator-order C operand
    def job(self, frags, cin):
        acc = cin.copy()
        for fr in frags:
            a = self.stream[self.pos * QUAD_ELEMS:(self.pos + 1) * QUAD_ELEMS].reshape(64, 8)
            self.pos += 1
            A = np.zeros((16, 32))
            B = np.zeros((32, 16))
            for q in range(4):
                A[:, 8 * q:8 * q + 8] = a[16 * q:16 * q + 16, :]
                B[8 * q:8 * q + 8, :] = fr[16 * q:16 * q + 16, :].T
            Dm = A @ B                                                   # [row, col]
            for r in range(4):
                acc[r] += Dm[4 * Q4 + r, COL]
        return acc

    @staticmethod
    def cin_from(vec, t):
        c = np.zeros((4, 64))
        for r in range(4):
            c[r] = vec[MT * t + 4 * Q4 + r]
        return c

    @staticmethod
    def pack(acc0, acc1, relu):
        """accumulators of tiles 2s, 2s+1 ([4][64] each) -> fragment s [64][8] (bf16-rounded values)."""
        v = np.concatenate([acc0, acc1], 0)                              # elements 0..3 | 4..7
        v = np.maximum(v, 0) if relu else v
        return bf16_round(v.astype(np.float32)).astype(np.float64).T.copy()

    def layer(self, frags, bias, n_tiles, relu):
        accs = [self.job(frags, self.cin_from(bias, t)) for t in range(n_tiles)]
        return [self.pack(accs[2 * s], accs[2 * s + 1], relu) for s in range(n_tiles // 2)]

    def enc_frags(self, pts, sin_fn=np.sin, cos_fn=np.cos):
        """pts [16, 3] -> KPE fragments of gamma(x): slot u = 32 ks + 8 q + j is channel u."""
        L = self.L_x
        nch = 3 + 6 * L
        chan = np.zeros((16, ((nch + KF - 1) // KF) * KF))
        chan[:, :3] = pts
        for k in range(L):
            chan[:, 3 + 6 * k:3 + 6 * k + 3] = sin_fn(pts * 2.0 ** k)
            chan[:, 3 + 6 * k + 3:3 + 6 * k + 6] = cos_fn(pts * 2.0 ** k)
        chan = bf16_round(chan.astype(np.float32)).astype(np.float64)
        frags = []
        for ks in range(chan.shape[1] // KF):
            fr = np.zeros((64, 8))
            for q in range(4):
                fr[16 * q:16 * q + 16, :] = chan[:, KF * ks + 8 * q:KF * ks + 8 * q + 8]
            frags.append(fr)
        return frags

    def tile(self, pts, dir_gamma):
        """pts [16,3] sample positions of one 16-point tile, dir_gamma [in_d] = gamma(d/|d|) of its ray -> raw [16, 4]."""
        D, W, o, side = self.D, self.W, self.off, self.side
        NT = W // MT
        skip_layer = self.skip + 1 if (self.skip >= 0 and self.skip + 1 < D) else -1
        self.pos = 0
        pe = self.enc_frags(pts)
        hb = self.layer(pe, side[o["bias_trunk"]:], NT, True)            # layer 0
        for l in range(1, D):
            hb = self.layer(hb + (pe if l == skip_layer else []), side[o["bias_trunk"] + l * W:], NT, True)   # activations first, gamma(x) last
        assert self.pos * 1024 == len(self.stream) * 2 - 224 * 1024
        feat = self.layer(hb, side[o["bias_feat"]:], NT, False)
        chead = np.zeros((4, 64))
        chead[3, :16] = side[o["head_b"] + 3]
        dens = self.job(hb, chead)                                       # density row over the trunk output
        in_d = 3 + 6 * self.L_d
        wdt = side[o["wdir_t"]:o["wdir_t"] + in_d * (W // 2)].reshape(in_d, W // 2)
        dbias = side[o["bias_d"]:o["bias_d"] + W // 2] + dir_gamma @ wdt
        g = self.layer(feat, dbias, NT // 2, True)
        ccol = np.zeros((4, 64))
        for r in range(3):
            ccol[r, :16] = side[o["head_b"] + r]
        colour = self.job(g, ccol)                                       # colour rows over the view-direction output
        self.pos += 20                                                   # padding
        assert self.pos == self.n_quads, (self.pos, self.n_quads)
        return np.stack([colour[0, :16], colour[1, :16], colour[2, :16], dens[3, :16]], 1)      # lanes 0..15
