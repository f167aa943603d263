"""CPU model of the lane-pair trade of the one-row-per-wave row kernels above 1024 (psfmc_amd/csrc/psfmc_rows3_path.h,
k_rows3_fwd / k_rows3_inv with kPair): layout [kx][y][c], lanes 2j / 2j + 1 hold kx_e / kx_e + 1, one value changes
hands so that each of the pair's two store (load) instructions moves the 32 adjacent bytes (kx; c = 0, c = 1), the
component = the lane's parity.  The model walks the kernel's own index arithmetic (fft3g_index / fft3g_valid, the
register-inclusion test, the store predicates) for the four built sides and checks that every (kx <= nx / 2, c) is
written exactly once, by the right value, and that the two lanes of a pair always address one sector -- including the
Nyquist column (held by lane 0 alone, stored with lane 1's help) and the sides whose last output block is only partly
filled (1152, 1280: R1 R2 is not a multiple of 64).  No GPU, no oracle: the device result itself is checked against the
oracle by tests/test_gpu_random.py."""
import itertools

import pytest

SIDES = {1152: 18, 1280: 20, 1536: 24, 2048: 32}      # nx: R1 (R2 = R3 = 8, L = 64)
EL = 16                                                # bytes of a T element


def lanes_of(nx, r1):
    r1r2 = r1 * 8
    nb3 = (r1r2 + 63) // 64
    return r1r2, nb3


def forward_stores(nx, r1):
    """[(instruction id, lane, byte offset inside the row's [kx][c] plane, value tag)] as the kernel issues them;
    value tag = ('G' | 'H', kx) -- what the lane holds after the trade."""
    r1r2, nb3 = lanes_of(nx, r1)
    out = []
    for q, k3 in itertools.product(range(nb3), range(8)):
        k0 = 64 * q + r1r2 * k3                       # lane 0's k of this register
        if 2 * k0 > nx:                                # (folds at compile time: 2 (k - t) <= NX)
            continue
        own = {}
        for t in range(64):
            k = t + k0
            held = t + 64 * q < r1r2
            own[t] = (held, k)
        for t in range(64):
            held, k = own[t]
            odd = t & 1
            mate_held, mate_k = own[t ^ 1]
            assert held == mate_held, 'the lanes of a pair hold outputs together'
            # the even lane gives H[k_e] and takes G[k_e + 1]; the odd lane gives G[k_o] and takes H[k_o - 1]
            got = ('G', mate_k) if not odd else ('H', mate_k)
            first = ('G', k) if not odd else got
            second = got if not odd else ('H', k)
            ke, ko = k & ~1, k | 1
            ins = (q * 8 + k3) * 2
            if held and 2 * ke <= nx:
                out.append((ins, t, (ke * 2 + odd) * EL, first))
            if held and 2 * ko <= nx:
                out.append((ins + 1, t, (ko * 2 + odd) * EL, second))
    return out


@pytest.mark.parametrize('nx', sorted(SIDES))
def test_forward_pair_stores_cover_every_element_once(nx):
    stores = forward_stores(nx, SIDES[nx])
    written = {}
    for ins, t, off, tag in stores:
        assert off not in written, (nx, off)
        written[off] = tag
    want = {}
    for kx in range(nx // 2 + 1):
        want[(kx * 2 + 0) * EL] = ('G', kx)
        want[(kx * 2 + 1) * EL] = ('H', kx)
    assert written == want
    # the two lanes of a pair fill one 32-byte sector in one instruction
    by_ins = {}
    for ins, t, off, tag in stores:
        by_ins.setdefault((ins, t >> 1), []).append(off)
    for (ins, pair), offs in by_ins.items():
        assert len(offs) == 2 and abs(offs[0] - offs[1]) == EL and min(offs) % (2 * EL) == 0, (nx, ins, pair, offs)


@pytest.mark.parametrize('nx', sorted(SIDES))
def test_inverse_pair_loads_hand_every_lane_its_own_g_and_h(nx):
    r1 = SIDES[nx]
    for a in range(r1):
        if 2 * 64 * a > nx:                            # (folds) no lane of this register is in the lower half
            continue
        loaded = {}
        for t in range(64):
            k = 64 * a + t
            odd = t & 1
            ke, ko = k & ~1, k | 1
            # first / second = what the lane's two load instructions fetch: (k_e, c = parity), (k_o, c = parity)
            first = (('G', 'H')[odd], ke) if 2 * ke <= nx else None
            second = (('G', 'H')[odd], ko) if 2 * ko <= nx else None
            loaded[t] = (first, second)
        for t in range(64):
            k = 64 * a + t
            odd = t & 1
            first, second = loaded[t]
            m_first, m_second = loaded[t ^ 1]
            # the kernel: got = pair_trade(odd, second, first) -- the even lane gives its `second`, the odd lane its
            # `first`, and each takes what the other gave
            got = m_first if not odd else m_second
            g = first if not odd else got
            h = got if not odd else second
            if 2 * k <= nx:
                assert g == ('G', k) and h == ('H', k), (nx, a, t, g, h)


@pytest.mark.parametrize('nx', [1152, 2048])
def test_half_stage1_table_product_is_the_missing_half(nx):
    """fft_wave3g HALF1 (psfmc_fft.h): the workgroup's stage-1 table holds W^(t k1) for k1 < R1 / 2 only and the other
    half is formed as W^(t k1) W^(t R1 / 2).  With table entries rounded from long double (as the host computes them)
    the product is within 2 ulp of the directly rounded entry -- the transforms above 1024 carry that much extra
    twiddle error in half of their stage-1 factors."""
    import numpy as np
    r1 = SIDES[nx]
    j = np.arange(nx, dtype=np.longdouble)
    ang = -2.0 * np.pi * j / nx
    tw = (np.cos(ang).astype(np.float64) + 1j * np.sin(ang).astype(np.float64))       # exp(-2 pi i j / nx), rounded once
    h = r1 // 2
    t = np.arange(64)[:, None]
    k = np.arange(h)[None, :]
    direct = tw[(t * (k + h)) % nx]
    formed = tw[(t * k) % nx] * tw[(t * h) % nx]
    assert np.max(np.abs(formed - direct)) <= 2.5 * np.finfo(np.float64).eps
