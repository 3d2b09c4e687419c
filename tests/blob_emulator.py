"""Lane-level numpy emulation of mlp_fp32.hip's dataflow, driven by a packed blob.

Checks the host packer (pack.cpp) and the kernel's operand bookkeeping without a GPU: it walks the
weight stream quad by quad exactly as the kernel does, applies v_mfma_f32_32x32x2_f32 semantics
(A[i][k]: lane = i + 32k;  B[k][j]: lane = j + 32k;  D[i][j]: lane = j + 32*((i>>2)&1), reg = (i&3) + 4*(i>>3))
and the same epilogues.  fp64 accumulation: this is a layout check, not a rounding model.
"""
import numpy as np

QUAD = 256  # floats


def _hdr(blob):
    h = np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)
    assert h[0] == 0x4D494E46
    return dict(D=int(h[2]), W=int(h[3]), skip=int(np.int32(h[4])), L_x=int(h[5]), L_d=int(h[6]), stream_off=int(h[7]),
                hoist=int(h[8]), full=int(h[9]), side_off=int(h[10]), side_floats=int(h[11]))


def _side_offsets(D, W, in_d):
    o, f = {}, 0
    for name, n in (("bias_trunk", D * W), ("bias_feat", W), ("bias_d", W // 2), ("dens_w", W), ("dens_b", 4),
                    ("color_w", 3 * (W // 2)), ("color_b", 4), ("wdir_t", in_d * (W // 2))):
        o[name] = f
        f += n
    return o


LANE = np.arange(64)
COL, HH = LANE & 31, LANE >> 5


def _row_of(r, hh):
    return (r & 3) + 8 * (r >> 2) + 4 * hh


class Emu:
    def __init__(self, blob: np.ndarray):
        self.h = _hdr(blob)
        h = self.h
        self.stream = np.frombuffer(blob[h["stream_off"]:h["stream_off"] + h["full"]].tobytes(), dtype=np.float32).astype(np.float64)
        self.side = np.frombuffer(blob[h["side_off"]:h["side_off"] + 4 * h["side_floats"]].tobytes(), dtype=np.float32).astype(np.float64)
        self.off = _side_offsets(h["D"], h["W"], 3 + 6 * h["L_d"])
        self.pos = 0  # quad cursor

    # --- kernel building blocks -------------------------------------------------------------
    def acc_init(self, NT, vec):
        acc = np.zeros((NT, 16, 64))
        for t in range(NT):
            for r in range(16):
                acc[t, r] = vec[32 * t + _row_of(r, HH)]
        return acc

    def gemm_part(self, acc, NT, b):
        """b: [KS, 64] per-lane B registers."""
        KS = b.shape[0]
        start = self.pos
        for kq in range(KS // 4):
            for t in range(NT):
                a = self.stream[self.pos * QUAD:(self.pos + 1) * QUAD].reshape(64, 4)
                self.pos += 1
                for j in range(4):
                    A = np.stack([a[:32, j], a[32:, j]], 1)                 # [i, k]
                    B = np.stack([b[4 * kq + j][:32], b[4 * kq + j][32:]], 0)  # [k, jcol]
                    Dm = A @ B                                              # [i, jcol]
                    for r in range(16):
                        acc[t, r] += Dm[_row_of(r, HH), COL]
        used = self.pos - start
        self.pos = start + (used + 15) // 16 * 16                           # parts are slot aligned
        return acc

    @staticmethod
    def acc_to_b(acc, NT, relu):
        h = acc[:NT].reshape(NT * 16, 64).copy()
        return np.maximum(h, 0) if relu else h

    def dot_half(self, h, w):
        n = h.shape[0]
        s = np.zeros(64)
        for q in range(n // 4):
            for e in range(4):
                s += h[4 * q + e] * w[8 * q + 4 * HH + e]
        return s[:32] + s[32:]                                               # xhalf_sum, per point

    @staticmethod
    def enc_regs(L, p, sin_fn=np.sin, cos_fn=np.cos):
        """p [3, 32] per-point coords -> [pe_ksteps, 64] registers."""
        K = ((3 * L + 2 + 3) // 4) * 4
        pe = np.zeros((K, 64))
        for s in range(3 * L):
            y = p[s % 3] * float(1 << (s // 3))
            pe[s, :32], pe[s, 32:] = sin_fn(y), cos_fn(y)
        pe[3 * L, :32], pe[3 * L, 32:] = p[0], p[1]
        pe[3 * L + 1, :32] = p[2]
        return pe

    # --- one 32-point tile --------------------------------------------------------------------
    def tile(self, pe, de=None, dir_gamma=None):
        """pe [KPE,64] encoded position regs; either de [KDE,64] (embedded mode) or dir_gamma [in_d] (hoisted)."""
        h_, o, side = self.h, self.off, self.side
        D, W = h_["D"], h_["W"]
        NT, HN = W // 32, W // 2
        skip_layer = h_["skip"] + 1 if (h_["skip"] >= 0 and h_["skip"] + 1 < D) else -1
        self.pos = 0
        acc = self.acc_init(NT, side[o["bias_trunk"]:])
        acc = self.gemm_part(acc, NT, pe)
        for l in range(1, D):
            h = self.acc_to_b(acc, NT, True)
            acc = self.acc_init(NT, side[o["bias_trunk"] + l * W:])
            if l == skip_layer:
                acc = self.gemm_part(acc, NT, pe)
            acc = self.gemm_part(acc, NT, h)
        h = self.acc_to_b(acc, NT, True)
        dens = self.dot_half(h, side[o["dens_w"]:]) + side[o["dens_b"]]
        acc = self.acc_init(NT, side[o["bias_feat"]:])
        acc = self.gemm_part(acc, NT, h)
        h = self.acc_to_b(acc, NT, False)
        if de is None:
            in_d = 3 + 6 * h_["L_d"]
            wdt = side[o["wdir_t"]:o["wdir_t"] + in_d * (W // 2)].reshape(in_d, W // 2)
            dbias = side[o["bias_d"]:o["bias_d"] + W // 2] + dir_gamma @ wdt
            acc = self.acc_init(NT // 2, dbias)
            acc = self.gemm_part(acc, NT // 2, h)
            assert self.pos * QUAD * 4 == h_["hoist"]
        else:
            acc = self.acc_init(NT // 2, side[o["bias_d"]:])
            acc = self.gemm_part(acc, NT // 2, h)
            acc = self.gemm_part(acc, NT // 2, de)
            assert self.pos * QUAD * 4 == h_["full"]
        h2 = self.acc_to_b(acc, NT // 2, True)
        cw = side[o["color_w"]:]
        rgb = [self.dot_half(h2, cw[c * (W // 2):]) + side[o["color_b"] + c] for c in range(3)]
        return np.stack(rgb + [dens], 1)                                    # [32, 4]


# ------------------------------------------------------------------------------------------------------------
# mlp_fp32_wide.hip: 16 points per wave on v_mfma_f32_16x16x4_f32 (networks wider than 256, packed for W = 512 in the W16 stream order).
# A[i][k]: lane = i + 16k;  B[k][j]: lane = j + 16k;  D[i][j]: lane = j + 16*(i >> 2), reg = i & 3.
# ------------------------------------------------------------------------------------------------------------
COL16, Q4 = LANE & 15, LANE >> 4


class EmuWide:
    def __init__(self, blob: np.ndarray):
        self.h = _hdr(blob)
        h = self.h
        assert np.frombuffer(blob[:64].tobytes(), dtype=np.uint32)[1] == 6 and h["W"] in (384, 512)   # the W16 stream order
        self.stream = np.frombuffer(blob[h["stream_off"]:h["stream_off"] + h["full"]].tobytes(), dtype=np.float32).astype(np.float64)
        self.side = np.frombuffer(blob[h["side_off"]:h["side_off"] + 4 * h["side_floats"]].tobytes(), dtype=np.float32).astype(np.float64)
        self.off = _side_offsets(h["D"], h["W"], 3 + 6 * h["L_d"])
        self.pos = 0

    def acc_init(self, NT, vec):
        acc = np.zeros((NT, 4, 64))
        for t in range(NT):
            for r in range(4):
                acc[t, r] = vec[16 * t + 4 * Q4 + r]
        return acc

    def gemm_part(self, acc, NT, b):
        KS = b.shape[0]
        start = self.pos
        for kq in range(KS // 4):
            for t in range(NT):
                a = self.stream[self.pos * QUAD:(self.pos + 1) * QUAD].reshape(64, 4)
                self.pos += 1
                for j in range(4):
                    A = a[:, j].reshape(4, 16).T                            # [i, k]: lane = i + 16k
                    B = b[4 * kq + j].reshape(4, 16)                        # [k, jcol]
                    Dm = A @ B
                    for r in range(4):
                        acc[t, r] += Dm[4 * Q4 + r, COL16]
        used = self.pos - start
        assert used % 8 == 0                                                # the kernel's rotating file of 8 A quads
        self.pos = start + (used + 15) // 16 * 16                           # parts are slot aligned (W = 384: the 24-quad direction part)
        return acc

    @staticmethod
    def acc_to_b(acc, NT, relu):
        h = acc[:NT].reshape(NT * 4, 64).copy()
        return np.maximum(h, 0) if relu else h

    def dot_quarter(self, h, w):
        s = np.zeros(64)
        for t in range(h.shape[0] // 4):
            for r in range(4):
                s += h[4 * t + r] * w[16 * t + 4 * Q4 + r]
        return s.reshape(4, 16).sum(0)                                       # quarter_sum, per point

    @staticmethod
    def steps(L):
        return (((3 * L + 1) // 2 + 1 + 3) // 4) * 4

    @classmethod
    def enc_regs(cls, L, p):
        """p [3, 16] per-point coords -> [pe_ksteps16, 64] registers."""
        pe = np.zeros((cls.steps(L), 64))
        n = (3 * L + 1) // 2
        for s in range(n):
            for q in range(4):
                m = 2 * s + (q >> 1)
                if m < 3 * L:
                    y = p[m % 3] * float(1 << (m // 3))
                    pe[s, 16 * q:16 * q + 16] = np.cos(y) if (q & 1) else np.sin(y)
        for q in range(3):
            pe[n, 16 * q:16 * q + 16] = p[q]
        return pe

    @classmethod
    def gather_regs(cls, L, rows):
        """rows [16, 3 + 6L] pre-embedded -> the same registers (embedded mode)."""
        pe = np.zeros((cls.steps(L), 64))
        n = (3 * L + 1) // 2
        for s in range(n):
            for q in range(4):
                m = 2 * s + (q >> 1)
                if m < 3 * L:
                    pe[s, 16 * q:16 * q + 16] = rows[:, 3 + 6 * (m // 3) + (m % 3) + 3 * (q & 1)]
        for q in range(3):
            pe[n, 16 * q:16 * q + 16] = rows[:, q]
        return pe

    def tile(self, pe, de=None, dir_gamma=None):
        h_, o, side = self.h, self.off, self.side
        D, W = h_["D"], h_["W"]
        NT, HN = W // 16, W // 4
        skip_layer = h_["skip"] + 1 if (h_["skip"] >= 0 and h_["skip"] + 1 < D) else -1
        self.pos = 0
        acc = self.gemm_part(self.acc_init(NT, side[o["bias_trunk"]:]), NT, pe)
        for l in range(1, D):
            h = self.acc_to_b(acc, NT, True)
            acc = self.acc_init(NT, side[o["bias_trunk"] + l * W:])
            if l == skip_layer:
                acc = self.gemm_part(acc, NT, pe)
            acc = self.gemm_part(acc, NT, h)
        h = self.acc_to_b(acc, NT, True)
        dens = self.dot_quarter(h, side[o["dens_w"]:]) + side[o["dens_b"]]
        acc = self.gemm_part(self.acc_init(NT, side[o["bias_feat"]:]), NT, h)
        h = self.acc_to_b(acc, NT, False)
        if de is None:
            in_d = 3 + 6 * h_["L_d"]
            wdt = side[o["wdir_t"]:o["wdir_t"] + in_d * (W // 2)].reshape(in_d, W // 2)
            acc = self.gemm_part(self.acc_init(NT // 2, side[o["bias_d"]:o["bias_d"] + W // 2] + dir_gamma @ wdt), NT // 2, h)
            assert self.pos * QUAD * 4 == h_["hoist"]
        else:
            acc = self.gemm_part(self.acc_init(NT // 2, side[o["bias_d"]:]), NT // 2, h)
            acc = self.gemm_part(acc, NT // 2, de)
            assert self.pos * QUAD * 4 == h_["full"]
        h2 = self.acc_to_b(acc, NT // 2, True)
        cw = side[o["color_w"]:]
        rgb = [self.dot_quarter(h2, cw[c * (W // 2):]) + side[o["color_b"] + c] for c in range(3)]
        return np.stack(rgb + [dens], 1)                                    # [16, 4]
