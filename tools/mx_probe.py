"""(GPU box) Pins the lane maps and the scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with FP8 (e4m3) operands through the raw-register
probe of the diagnostics library (svps_probe_mx_fp8), before any kernel relies on them:
  1. which (lane half, byte) of the A operand multiplies which (lane half, byte) of the B operand (one-hot k on both sides: 64 x 64 problems)
  2. which lane supplies which row of A / column of B (one-hot lane)
  3. what the scale operands do (E8M0 in byte 0: x 2^(e - 127)?), and whether the other bytes matter
    python tools/mx_probe.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotvps_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load_diag()
ONE, TWO = 0x38, 0x40                      # e4m3: 1.0, 2.0


def run(a_bytes, b_bytes, sa, sb):
    """a_bytes, b_bytes [n, 64, 32] uint8; sa, sb [n, 64] int32 -> C [n, 64, 16] fp32"""
    n = a_bytes.shape[0]
    a = torch.from_numpy(np.ascontiguousarray(a_bytes)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(b_bytes)).to(dev)
    s_a = torch.from_numpy(np.ascontiguousarray(sa, dtype=np.int32)).to(dev)
    s_b = torch.from_numpy(np.ascontiguousarray(sb, dtype=np.int32)).to(dev)
    c = torch.zeros((n, 64, 16), dtype=torch.float32, device=dev)
    rc = lib.svps_probe_mx_fp8(ops._ptr(a), ops._ptr(b), ops._ptr(s_a), ops._ptr(s_b), ops._ptr(c), n, ops._stream_ptr(dev))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return c.cpu().numpy()


lanes = np.arange(64)
# ---- 1. k pairing
n = 64 * 64
A = np.zeros((n, 64, 32), np.uint8)
B = np.zeros((n, 64, 32), np.uint8)
for ia in range(64):
    ha, pa = divmod(ia, 32)
    for ib in range(64):
        hb, pb = divmod(ib, 32)
        i = ia * 64 + ib
        A[i, lanes[(lanes >> 5) == ha], pa] = ONE
        B[i, lanes[(lanes >> 5) == hb], pb] = ONE
S = np.full((n, 64), 127, np.int32)
C = run(A, B, S, S)
M = C[:, 0, 0].reshape(64, 64)
assert set(np.unique(M)) <= {0.0, 1.0}, np.unique(M)
pair = M.argmax(1)
print("k pairing is a permutation:", bool((M.sum(1) == 1).all() and (M.sum(0) == 1).all()), "| identity (same lane half, same byte):", bool((pair == np.arange(64)).all()))
if not (pair == np.arange(64)).all():
    print("A (half, byte) -> B (half, byte):", [(divmod(i, 32), divmod(int(p), 32)) for i, p in enumerate(pair) if p != i][:16])
assert (C[:, :, :] == C[:, :1, :1]).all(), "with all rows / columns equal every result element must be the same"

# ---- 2. rows of A / columns of B: one-hot lane
A = np.zeros((64, 64, 32), np.uint8)
B = np.full((64, 64, 32), ONE, np.uint8)
for l in range(64):
    A[l, l, :] = ONE
C = run(A, B, S[:64], S[:64])
ok_rows = True
for l in range(64):
    nz = np.argwhere(C[l] != 0)                      # (lane, reg) of the nonzero results
    rows = {(int(reg) & 3) + 8 * (int(reg) >> 2) + 4 * (int(ln) >> 5) for ln, reg in nz}
    ok_rows &= rows == {l & 31} and len(nz) == 32 and np.allclose(C[l][C[l] != 0], 32.0)
print("A: lane l supplies row l & 31, k-half l >> 5 (32 k values each):", ok_rows)
A = np.full((64, 64, 32), ONE, np.uint8)
B = np.zeros((64, 64, 32), np.uint8)
for l in range(64):
    B[l, l, :] = ONE
C = run(A, B, S[:64], S[:64])
ok_cols = all({int(ln) & 31 for ln, reg in np.argwhere(C[l] != 0)} == {l & 31} and (C[l] != 0).sum() == 32 for l in range(64))
print("B: lane l supplies column l & 31, k-half l >> 5:", ok_cols)

# ---- 3. scales
A = np.full((6, 64, 32), ONE, np.uint8)
B = np.full((6, 64, 32), ONE, np.uint8)
sa = np.full((6, 64), 127, np.int32)
sb = np.full((6, 64), 127, np.int32)
sa[1] = 128                                  # A x 2
sb[2] = 125                                  # B / 4
sa[3] = 127 | (0x55 << 8) | (0x33 << 16) | (0x7f << 24)   # other bytes set
sa[4, :32] = 129                             # per lane: rows of the first k-half x 4
sb[5, 3] = 130                               # one lane of B: column 3, first k-half x 8
C = run(A, B, sa, sb)
print("scales 127/127 ->", C[0, 0, 0], "| A 128 ->", C[1, 0, 0], "| B 125 ->", C[2, 0, 0], "| A 127 with other bytes set ->", C[3, 0, 0],
      "| A lanes 0-31 at 129 ->", C[4, 0, 0], "| B lane 3 at 130: column 3 ->", C[5, 3, 0], "other columns ->", C[5, 4, 0])


# ---- 4. v_cvt_scalef32_pk_fp8_f16: multiply or divide by the scale, rounding, saturation
def e4m3_decode(b):
    b = b.astype(np.int64)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = (b >> 3) & 0xf
    m = b & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    return np.where((b & 0x7f) == 0x7f, np.nan, s * v)


vals = np.concatenate([np.linspace(-20, 20, 2001), [0.001, 0.01, 0.07, 100.0, 300.0, 447.0, 449.0, 500.0, 1000.0, 60000.0, -60000.0, 1e-4]]).astype(np.float16)
if vals.size % 2:
    vals = np.concatenate([vals, [np.float16(0)]])
xp = torch.from_numpy(vals.view(np.int16).copy()).to(dev)
for scale in (1.0, 4.0, 0.25):
    out = torch.zeros(vals.size // 2, dtype=torch.int32, device=dev)
    rc = lib.svps_probe_cvt_fp8(ops._ptr(xp), float(scale), ops._ptr(out), vals.size // 2, ops._stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    got = np.stack([e4m3_decode(o & 0xff), e4m3_decode((o >> 8) & 0xff)], 1).reshape(-1)
    v = vals.astype(np.float64)
    small = np.abs(v) < 100
    err_mul = np.nanmax(np.abs(got[small] - v[small] * scale) / np.maximum(np.abs(v[small] * scale), 2.0 ** -6))
    err_div = np.nanmax(np.abs(got[small] - v[small] / scale) / np.maximum(np.abs(v[small] / scale), 2.0 ** -6))
    print(f"scale {scale}: max rel err if out = x * scale: {err_mul:.3f}; if out = x / scale: {err_div:.3f}; 60000 -> {got[-3]}, -60000 -> {got[-2]}, 1000 -> {got[-4]}, 449 -> {got[-6]}")


# ---- 5. the building block of the F8 form of retr_stats_hl.hip end to end: fp16 operands -> in-kernel FP8 conversion -> scaled MFMA
g = np.random.default_rng(0)
a = (g.standard_normal((32, 64)) * 3e-4).astype(np.float16)          # like R_lo
b = (g.standard_normal((32, 64)) * 2.0).astype(np.float16)           # like x_hi
def sbyte(x):
    am = np.abs(x.astype(np.float32)).max()
    return int((np.float32(am).view(np.uint32) >> 23) & 0xff) - 7
sa = np.zeros(64, np.int32); sb = np.zeros(64, np.int32)
for lane in range(64):
    r_, h_ = lane & 31, lane >> 5
    cols = np.concatenate([np.arange(16 * q + 8 * h_, 16 * q + 8 * h_ + 8) for q in range(4)])
    # ONE byte per row / column: the instruction's scale blocks (k = 0 .. 31: byte of lane r; k = 32 .. 63: byte of lane r + 32) are each spread
    # over both lanes of a row (a lane's dwords 0 - 3 / 4 - 7) - with different bytes in the two lanes of a row the result is NOT the per-lane
    # scaled sum (found with exactly this test)
    sa[lane] = sbyte(a[r_]); sb[lane] = sbyte(b[r_])
ta, tb = torch.from_numpy(a.view(np.int16).copy()).to(dev), torch.from_numpy(b.view(np.int16).copy()).to(dev)
tsa, tsb = torch.from_numpy(sa).to(dev), torch.from_numpy(sb).to(dev)
c = torch.zeros((64, 16), dtype=torch.float32, device=dev)
regs = torch.zeros((64, 16), dtype=torch.int32, device=dev)
rc = lib.svps_probe_mx_block(ops._ptr(ta), ops._ptr(tb), ops._ptr(tsa), ops._ptr(tsb), ops._ptr(c), ops._ptr(regs), ops._stream_ptr(dev))
assert rc == 0
torch.cuda.synchronize()
C = c.cpu().numpy(); R = regs.cpu().numpy().view(np.uint32)
want = a.astype(np.float64) @ b.astype(np.float64).T                # [row, col]
got = np.zeros((32, 32))
for lane in range(64):
    for reg in range(16):
        got[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), lane & 31] = C[lane, reg]
print("block product: max |got - want| / max |want| =", np.abs(got - want).max() / np.abs(want).max(), "(FP8 rounding of both operands: ~0.05 expected)")
# decode lane 0's A registers and compare with the source values
bytes0 = np.array([(R[0, i] >> (8 * k)) & 0xff for i in range(8) for k in range(4)], dtype=np.uint8)
dec = e4m3_decode(bytes0) * 2.0 ** (sa[0] - 127)
src = np.concatenate([a[0, 16 * q: 16 * q + 8] for q in range(4)]).astype(np.float64)
print("lane 0 A registers decoded vs source (first 8):", dec[:8], src[:8])
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mx_block_dump.npz"), C=C, R=R, sa=sa, sb=sb, a=a, b=b, got=got, want=want)
