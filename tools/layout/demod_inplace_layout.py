"""Not a test of the product: checks the index algebra of k_demod's in-place first exchange (fft2048.h: fft_round_a_inplace /
fft_rounds_bc_split) -- bijections, who-reads-what between barriers, and LDS bank conflicts by the rules of
MI355X_MICROARCH.md (ds_read_b64: groups of 32 lanes, bank = (a/4) mod 64; ds_write_b64: groups of 16 lanes, (a/4) mod 32).
Element = one cf32 (8 bytes).  Run: python tools/layout/demod_inplace_layout.py"""
import itertools

PITCH = 260                    # elements per row of 256 raw samples: 32 bytes of padding (two 16-byte DMA units)
HALF = 4 * PITCH               # element offset of the second wave's half of the tile
def A0(n):                     # element address of raw sample n / of the round-A output written over it
    return (n & 255) + PITCH * (n >> 8)
def CK(t):                     # round B's thread -> (c, k): c = 4 j1 + j2, lanes ordered k, j1, j2
    k, j1, j2 = t & 7, (t >> 3) & 3, t >> 5
    return 4 * j1 + j2, k
def E2(p):                     # element address of position p after round B: wave (p>>6)&1 of round C owns half a tile
    q = (p & 63) | ((p >> 7) << 6)
    return HALF * ((p >> 6) & 1) + (q ^ (((q >> 8) & 1) << 3))

def conflicts_read_b64(addrs):          # addrs[lane]; worst multiplicity over the two 32-lane groups
    worst = 1
    for g in range(2):
        banks = {}
        for a in addrs[32 * g:32 * g + 32]:
            banks.setdefault(a % 32, set()).add(a)      # element % 32 <=> dword bank pair of 64 banks
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def conflicts_write_b64(addrs):
    worst = 1
    for g in range(4):
        banks = {}
        for a in addrs[16 * g:16 * g + 16]:
            banks.setdefault(a % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst

assert len(set(A0(n) for n in range(2048))) == 2048 and max(A0(n) for n in range(2048)) < 8 * PITCH
assert len(set(E2(p) for p in range(2048))) == 2048 and max(E2(p) for p in range(2048)) < 8 * PITCH
# the DMA moves 16-byte units: a sample pair must stay together and 16-byte aligned
for m in range(1024):
    assert A0(2 * m) % 2 == 0 and A0(2 * m + 1) == A0(2 * m) + 1
# round A: thread t, half h reads samples t + 128h + 256j and writes its output r over sample t + 128h + 256r
wA = 1
for w in range(2):
    for h in range(2):
        for j in range(8):
            a = [A0(t + 128 * h + 256 * j) for t in range(64 * w, 64 * w + 64)]
            wA = max(wA, conflicts_read_b64(a), conflicts_write_b64(a))
print("round A reads / in-place writes: worst", wA, "-way")
# position p = 8q + r is output r of thread-half b(q): q = 64 j1 + 16 j2 + 4 j3 + j4, b = j1 + 4 j2 + 16 j3 + 64 j4
def where_after_a(p):
    q, r = p >> 3, p & 7
    j1, j2, j3, j4 = q >> 6, (q >> 4) & 3, (q >> 2) & 3, q & 3
    b = j1 + 4 * j2 + 16 * j3 + 64 * j4
    return A0(b + 256 * r)
assert len(set(where_after_a(p) for p in range(2048))) == 2048
# round B: thread t -> (c, k) = CK(t) reads positions 128c + k + 8a + 32b
wB = 1
for w in range(2):
    for a_, b_ in itertools.product(range(4), range(4)):
        ad = []
        for t in range(64 * w, 64 * w + 64):
            c, k = CK(t)
            p = 128 * c + k + 8 * a_ + 32 * b_
            e = (c >> 2) + 4 * (c & 3) + PITCH * k + 16 * b_ + 64 * a_            # the closed form the kernel uses
            assert e == where_after_a(p)
            ad.append(e)
        wB = max(wB, conflicts_read_b64(ad))
print("round B reads: worst", wB, "-way")
wE = 1; wC = 1
for w in range(2):
    for a_, b_ in itertools.product(range(4), range(4)):
        ad = []
        for t in range(64 * w, 64 * w + 64):
            c, k = CK(t)
            e = HALF * (b_ >> 1) + 32 * (b_ & 1) + k + 64 * c + ((8 * a_) ^ (8 * ((c >> 2) & 1)))   # closed form of the kernel
            assert e == E2(128 * c + k + 8 * a_ + 32 * b_)
            ad.append(e)
        wE = max(wE, conflicts_write_b64(ad))
        ad = [E2(t + 128 * a_ + 512 * b_) for t in range(64 * w, 64 * w + 64)]
        assert ad == [HALF * (t >> 6) + ((t & 63) ^ (8 * (b_ & 1))) + 64 * a_ + 256 * b_ for t in range(64 * w, 64 * w + 64)]
        assert all(HALF * w <= x < HALF * w + 1024 for x in ad)          # a wave's round-C reads stay in its own half of the tile ...
        wC = max(wC, conflicts_read_b64(ad))
print("exchange 2 writes: worst", wE, "-way; round C reads: worst", wC, "-way")
# ... and so do the units its DMA fills (wave w copies samples 1024w .. 1024w + 1023)
for w in range(2):
    for n in range(1024 * w, 1024 * w + 1024):
        assert HALF * w <= A0(n) < HALF * (w + 1)
print("ok: a wave's DMA of the next symbol only overwrites elements that only this wave reads in round C")
