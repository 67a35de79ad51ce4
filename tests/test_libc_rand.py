"""The process-global libc rand() stream as the library sees it (slicer_rand.hip): shot-noise thinning
(densitymaps.cpp:387-397) continues that stream on the device, so the state has to be read and installed exactly.
Host-only checks (no GPU): the model recurrence against real rand() calls, rewind through the installer, and the jump
tables against step-by-step draws (through a small numpy twin of the tables' construction)."""
import ctypes as C

import numpy as np
import pytest

from slicer_amd import _lib

L = _lib.load()
libc = C.CDLL(None)
libc.rand.restype = C.c_int
libc.srand.argtypes = [C.c_uint]


def model_draws(v, k):
    """x[n] = x[n-31] + x[n-3] over Z/2^32, rand() = x[n] >> 1; v = 31 words, oldest first (advanced in place)."""
    v = [int(x) for x in v]
    out = []
    for _ in range(k):
        x = (v[0] + v[28]) & 0xFFFFFFFF
        v = v[1:] + [x]
        out.append(x >> 1)
    return out, v


def get_state():
    v = (C.c_uint32 * 31)()
    assert L.slicer_libc_rand_state_get(v) == 0
    return list(v)


def test_supported_here():
    assert L.slicer_libc_rand_supported() == 1


@pytest.mark.parametrize("seed,burn", [(1, 0), (12345, 7), (0xFFFFFFFF, 1000), (0, 31)])
def test_state_read_predicts_rand_and_reading_does_not_consume(seed, burn):
    libc.srand(seed)
    for _ in range(burn):
        libc.rand()
    v = get_state()
    v_again = get_state()
    assert v == v_again
    want, _ = model_draws(v, 200)
    got = [libc.rand() for _ in range(200)]
    assert got == want


def test_installing_a_state_rewinds_and_jumps_the_stream():
    libc.srand(777)
    for _ in range(5):
        libc.rand()
    v0 = get_state()
    first = [libc.rand() for _ in range(100)]
    arr = (C.c_uint32 * 31)(*v0)
    assert L.slicer_libc_rand_state_set(arr) == 0       # rewind
    assert [libc.rand() for _ in range(100)] == first
    # jump: install the state 1000 draws further on, computed by the model
    _, v1000 = model_draws(v0, 1000)
    assert L.slicer_libc_rand_state_set((C.c_uint32 * 31)(*v1000)) == 0
    after = [libc.rand() for _ in range(10)]
    libc.srand(777)
    ref = [libc.rand() for _ in range(5 + 1000 + 10)]
    assert after == ref[-10:]


def test_srand_after_an_installed_state_behaves_like_a_fresh_srand():
    libc.srand(99)
    v = get_state()
    assert L.slicer_libc_rand_state_set((C.c_uint32 * 31)(*v)) == 0
    libc.srand(4242)
    a = [libc.rand() for _ in range(50)]
    libc.srand(4242)
    assert [libc.rand() for _ in range(50)] == a
    want, _ = model_draws(get_state(), 5)
    assert [libc.rand() for _ in range(5)] == want


def test_jump_matrices_agree_with_stepping():
    """A^k by repeated squaring (what the device tables hold) applied to a state equals k single draws."""
    A = np.zeros((31, 31), np.uint64)
    for i in range(30):
        A[i, i + 1] = 1
    A[30, 0] = 1
    A[30, 28] = 1
    M = 0xFFFFFFFF

    def mul(a, b):
        # 32-bit wrapping product of two matrices of 32-bit words, without overflowing uint64 partial sums
        out = np.zeros((31, 31), np.uint64)
        for k in range(31):
            out = (out + ((a[:, k:k + 1] * b[k:k + 1, :]) & M)) & M
        return out

    libc.srand(31337)
    v = np.array(get_state(), np.uint64)
    P = A.copy()
    for k in range(12):  # A^(2^k)
        _, stepped = model_draws(v, 1 << k)
        jumped = np.zeros(31, np.uint64)
        for j in range(31):
            jumped = (jumped + ((P[:, j] * v[j]) & M)) & M
        assert [int(x) for x in jumped] == stepped
        P = mul(P, P)


def test_another_generator_type_is_reported_and_left_alone():
    """A process that switched libc to another generator (initstate with a 256-byte array: TYPE_4, degree 63) cannot have
    its stream continued by the 31-word model: the library says so (thinning then draws with rand() on the host) and does
    not disturb that generator."""
    libc.initstate.restype = C.c_void_p
    libc.initstate.argtypes = [C.c_uint, C.c_void_p, C.c_size_t]
    libc.setstate.restype = C.c_void_p
    libc.setstate.argtypes = [C.c_void_p]
    big = C.create_string_buffer(256)
    old = libc.initstate(4242, big, 256)
    try:
        a = [libc.rand() for _ in range(5)]
        assert L.slicer_libc_rand_supported() == 0
        v = (C.c_uint32 * 31)()
        assert L.slicer_libc_rand_state_get(v) != 0
        b = [libc.rand() for _ in range(5)]
        libc.initstate(4242, big, 256)
        assert [libc.rand() for _ in range(10)] == a + b
    finally:
        libc.setstate(old)
    assert L.slicer_libc_rand_supported() == 1
