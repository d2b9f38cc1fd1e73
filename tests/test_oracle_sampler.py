"""Pin the third-party arithmetic the samplers depend on:
PCG32 (rand_pcg 0.3), Standard f32 (rand 0.8), SipHash-1-3 (std DefaultHasher),
permutation_element (stratified.rs:147-178).  None of it is under /root/reference,
so these are the published algorithms' own vectors / structural checks."""
import ctypes as C
import struct

import numpy as np

from yuki_amd import abi

MASK = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & MASK


def siphash_py(key, msg, c_rounds, d_rounds):
    """Bit-level SipHash-c-d straight from the SipHash paper."""
    k0, k1 = struct.unpack("<QQ", key)
    v0, v1, v2, v3 = k0 ^ 0x736F6D6570736575, k1 ^ 0x646F72616E646F6D, k0 ^ 0x6C7967656E657261, k1 ^ 0x7465646279746573

    def rnd():
        nonlocal v0, v1, v2, v3
        v0 = (v0 + v1) & MASK; v1 = _rotl(v1, 13); v1 ^= v0; v0 = _rotl(v0, 32)
        v2 = (v2 + v3) & MASK; v3 = _rotl(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & MASK; v3 = _rotl(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & MASK; v1 = _rotl(v1, 17); v1 ^= v2; v2 = _rotl(v2, 32)

    n = len(msg)
    for i in range(0, n - n % 8, 8):
        (m,) = struct.unpack("<Q", msg[i : i + 8])
        v3 ^= m
        for _ in range(c_rounds):
            rnd()
        v0 ^= m
    b = ((n & 0xFF) << 56) | int.from_bytes(msg[n - n % 8 :], "little")
    v3 ^= b
    for _ in range(c_rounds):
        rnd()
    v0 ^= b
    v2 ^= 0xFF
    for _ in range(d_rounds):
        rnd()
    return v0 ^ v1 ^ v2 ^ v3


def test_python_siphash_restatement_against_the_paper_vector():
    # SipHash-2-4 reference vector (Aumasson & Bernstein, Appendix A): key 00..0f, msg 00..0e
    assert siphash_py(bytes(range(16)), bytes(range(15)), 2, 4) == 0xA129CA6149BE45E5


def test_oracle_siphash13_zero_key(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for n in list(range(0, 33)) + [100, 255, 256, 257]:
        msg = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        buf = (C.c_uint8 * max(1, n)).from_buffer_copy(msg or b"\0")
        assert L.orc_siphash13(buf, n) == siphash_py(bytes(16), msg, 1, 3), n


def test_pcg32_reference_vector(oracle):
    """pcg32_srandom_r(42, 54) of the PCG C reference (pcg32-demo) == Pcg32::new(42, 54)."""
    out = np.zeros(6, dtype=np.uint32)
    oracle.lib().orc_pcg32_sequence(42, 54, 0, out.ctypes.data_as(C.c_void_p), 6)
    assert [hex(v) for v in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg32_advance_equals_stepping(oracle):
    L = oracle.lib()
    for delta in (0, 1, 2, 65536, 65536 * 63 + 5, (1 << 40) + 12345):
        if delta <= 70000 * 64:
            seq = np.zeros(delta + 4, dtype=np.uint32)
            L.orc_pcg32_sequence(0x73B9642E74AC471C, 0xDEADBEEF12345678, 0, seq.ctypes.data_as(C.c_void_p), len(seq))
            adv = np.zeros(4, dtype=np.uint32)
            L.orc_pcg32_sequence(0x73B9642E74AC471C, 0xDEADBEEF12345678, delta, adv.ctypes.data_as(C.c_void_p), 4)
            assert np.array_equal(seq[delta:], adv), delta
        else:  # advance(a) then advance(b) == advance(a+b) is covered by splitting
            a = np.zeros(2, dtype=np.uint32)
            L.orc_pcg32_sequence(1, 2, delta, a.ctypes.data_as(C.c_void_p), 2)
            assert a[0] != a[1]


def test_permutation_element_is_a_permutation(oracle):
    L = oracle.lib()
    for l in (1, 2, 3, 4, 9, 16, 63, 64, 100, 256):
        for p in (0, 1, 0xDEADBEEF, 0xFFFFFFFF, 12345):
            if (l & (l - 1)) and p + l > 0xFFFFFFFF:
                continue  # (i + p) wraps in u32: the final `% l` is then not a bijection (inherent to the algorithm)
            vals = sorted(L.orc_permutation_element(i, l, p) for i in range(l))
            assert vals == list(range(l)), (l, p)


def test_uniform_sampler_is_the_pixel_stream(oracle):
    """uniform.rs:72-95: stream = SipHash13(pixel), state advanced by index*65536;
    get_1d = (u32 >> 8) * 2^-24."""
    L = oracle.lib()
    seed = 0x73B9642E74AC471C
    s = abi.SamplerDesc(abi.SAMPLER_UNIFORM, 16, 1, 1, seed)
    px, py, idx = 123, 456, 5
    stream = siphash_py(bytes(16), struct.pack("<HH", px, py), 1, 3)
    raw = np.zeros(6, dtype=np.uint32)
    L.orc_pcg32_sequence(seed, stream, idx * 65536, raw.ctypes.data_as(C.c_void_p), 6)
    dims = np.array([2, 1, 2, 1], dtype=np.uint8)
    out = np.zeros((4, 2), dtype=np.float32)
    L.orc_sampler_sequence(C.byref(s), px, py, idx, dims.ctypes.data_as(C.c_void_p), 4, out.ctypes.data_as(C.c_void_p))
    f = (raw >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    assert np.array_equal(out.reshape(-1)[[0, 1, 2, 4, 5, 6]], f)
    assert (f >= 0).all() and (f < 1).all()


def test_stratified_sampler_strata(oracle):
    """stratified.rs:104-144: over the spp sample indices of a pixel every
    dimension visits each stratum exactly once (a permutation); 2-D cells are
    x = s % nx, y = s / ny (sic); without jitter the offset is 0.5."""
    L = oracle.lib()
    nx = ny = 4
    s = abi.SamplerDesc(abi.SAMPLER_STRATIFIED, nx, ny, 0, 0x1234)
    dims = np.array([2, 2, 1], dtype=np.uint8)
    cells = [set(), set(), set()]
    for idx in range(nx * ny):
        out = np.zeros((3, 2), dtype=np.float32)
        L.orc_sampler_sequence(C.byref(s), 7, 9, idx, dims.ctypes.data_as(C.c_void_p), 3, out.ctypes.data_as(C.c_void_p))
        for k in range(2):
            cx, cy = out[k, 0] * nx - 0.5, out[k, 1] * ny - 0.5
            assert abs(cx - round(cx)) < 1e-6 and abs(cy - round(cy)) < 1e-6
            cells[k].add((round(cx), round(cy)))
        c = out[2, 0] * (nx * ny) - 0.5
        assert abs(c - round(c)) < 1e-5
        cells[2].add(round(c))
    assert len(cells[0]) == nx * ny and len(cells[1]) == nx * ny and cells[2] == set(range(nx * ny))
    assert cells[0] != [] and cells[0] == cells[1]  # same cell set, different order per dimension
