"""Lane-level numpy emulation of mlp_bf16.hip's dataflow, driven by a packed bf16 blob.

Checks the host packer (pack_bf16 in mlp_bf16.hip) and the kernel's operand bookkeeping without a GPU: it walks the weight
stream quad by quad in the kernel's order (output-tile-major jobs, the head tile chained over two inputs, the padded tail),
applies v_mfma_f32_32x32x16_bf16 semantics
    A fragment: lane l (i = l&31, h = l>>5), element j -> A[i][k = 8h + j]
    B fragment: lane l, element j                      -> B[k = 8h + j][col = l&31]
    D:          lane l, register r                     -> D[row = (r&3) + 8(r>>2) + 4(l>>5)][col = l&31]
and the same packing (accumulator registers 8s..8s+7 of tile t -> fragment 2t+s, rounded to bf16, optional ReLU).
Accumulation is fp64: this is a layout check with the kernel's ROUNDING POINTS (bf16 weights, bf16 activations, bf16 encoded
inputs), not a model of fp32 summation order.
"""
import numpy as np

QUAD_ELEMS = 512        # bf16 elements per 1 KiB quad
LANE = np.arange(64)
COL, HH = LANE & 31, LANE >> 5


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, elementwise."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _row_of(r, hh):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


class EmuBf16:
    def __init__(self, blob: np.ndarray):
        h = np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)
        assert h[0] == 0x4D494E46 and h[12] == 2
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

    # one job: output tile (32 rows) over the k-steps of `frags` ([KS][64 lanes][8]); cin [16][64] accumulator-order C operand
    def job(self, frags, cin):
        acc = cin.copy()
        for fr in frags:
            a = self.stream[self.pos * QUAD_ELEMS:(self.pos + 1) * QUAD_ELEMS].reshape(64, 8)
            self.pos += 1
            A = np.zeros((32, 16))
            B = np.zeros((16, 32))
            for h in range(2):
                A[:, 8 * h:8 * h + 8] = a[32 * h:32 * h + 32, :]
                B[8 * h:8 * h + 8, :] = fr[32 * h:32 * h + 32, :].T
            Dm = A @ B                                                   # [row, col]
            for r in range(16):
                acc[r] += Dm[_row_of(r, HH), COL]
        return acc

    @staticmethod
    def cin_from(vec, t):
        c = np.zeros((16, 64))
        for r in range(16):
            c[r] = vec[32 * t + _row_of(r, HH)]
        return c

    @staticmethod
    def pack(acc, relu):
        """accumulator [16][64] -> two fragments [64][8] (bf16-rounded values)."""
        v = np.maximum(acc, 0) if relu else acc
        v = bf16_round(v.astype(np.float32)).astype(np.float64)
        return [v[8 * s:8 * s + 8].T.copy() for s in range(2)]

    def enc_frags(self, pts, sin_fn=np.sin, cos_fn=np.cos):
        """pts [32, 3] -> KPE fragments of gamma(x): slot u = 16 ks + 8 h + j is channel u."""
        L = self.L_x
        nch = 3 + 6 * L
        chan = np.zeros((32, ((nch + 15) // 16) * 16))
        chan[:, :3] = pts
        for k in range(L):
            chan[:, 3 + 6 * k:3 + 6 * k + 3] = sin_fn(pts * 2.0 ** k)
            chan[:, 3 + 6 * k + 3:3 + 6 * k + 6] = cos_fn(pts * 2.0 ** k)
        chan = bf16_round(chan.astype(np.float32)).astype(np.float64)
        frags = []
        for ks in range(chan.shape[1] // 16):
            fr = np.zeros((64, 8))
            for h in range(2):
                fr[32 * h:32 * h + 32, :] = chan[:, 16 * ks + 8 * h:16 * ks + 8 * h + 8]
            frags.append(fr)
        return frags

    def tile(self, pts, dir_gamma):
        """pts [32,3] sample positions of one 32-point tile, dir_gamma [in_d] = gamma(d/|d|) of its ray -> raw [32, 4]."""
        D, W, o, side = self.D, self.W, self.off, self.side
        NT = W // 32
        skip_layer = self.skip + 1 if (self.skip >= 0 and self.skip + 1 < D) else -1
        self.pos = 0
        pe = self.enc_frags(pts)
        hb = []
        for t in range(NT):                                              # layer 0
            hb += self.pack(self.job(pe, self.cin_from(side[o["bias_trunk"]:], t)), True)
        for l in range(1, D):
            nxt = []
            frs = hb + (pe if l == skip_layer else [])                   # activations first, gamma(x) last
            for t in range(NT):
                nxt += self.pack(self.job(frs, self.cin_from(side[o["bias_trunk"] + l * W:], t)), True)
            hb = nxt
        assert self.pos * 1024 == len(self.stream) * 2 - 224 * 1024
        feat = []
        for t in range(NT):
            feat += self.pack(self.job(hb, self.cin_from(side[o["bias_feat"]:], t)), False)
        chead = np.zeros((16, 64))
        for r in range(4):
            chead[r, :32] = side[o["head_b"] + r]
        head = self.job(hb, chead)                                       # density row over the trunk output
        in_d = 3 + 6 * self.L_d
        wdt = side[o["wdir_t"]:o["wdir_t"] + in_d * (W // 2)].reshape(in_d, W // 2)
        dbias = side[o["bias_d"]:o["bias_d"] + W // 2] + dir_gamma @ wdt
        g = []
        for t in range(NT // 2):
            g += self.pack(self.job(feat, self.cin_from(dbias, t)), True)
        head = self.job(g, head)                                         # colour rows over the view-direction output
        self.pos += 8                                                    # padding
        assert self.pos == self.n_quads, (self.pos, self.n_quads)
        return np.stack([head[r, :32] for r in range(4)], 1)             # rows 0..3 = r, g, b, density  (lanes 0..31)
