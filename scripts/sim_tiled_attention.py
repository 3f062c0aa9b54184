"""CPU simulation of lane efficiency for LDS-tiled variants of the sparse attention kernel:
mapping A = one query row per 8-lane group (8 rows of a wave advance together through key tiles),
mapping B = one row per wave (8 entries per step).  Uses the real pixel-run structure (T=4096, T_M=256, k=64, H=32)."""
import numpy as np
rng = np.random.default_rng(0)
T, T_M, k, H = 4096, 256, 64, 32
def row_entries(t):
    # per head: K_t/H pixels kept (random), each pixel -> keys [b_m, b_m+1)
    w = t + 1
    Kt = max(1, round(H * k * T_M / w)); Kt = min(Kt, H * T_M)
    npix = max(1, min(T_M, int(round(Kt / H))))       # per head (approx, uniform split)
    pix = np.sort(rng.choice(T_M, size=npix, replace=False))
    b = np.floor(np.arange(T_M + 1) * (w / T_M) + 0.5).astype(int)
    runs = [(b[m], b[m + 1]) for m in pix if b[m + 1] > b[m]]
    return runs
def sim(KB, U=4, rows=range(64, T, 97)):
    tot_e = 0; tot_slots = 0; ideal_units = 0
    for t0 in rows:
        t0 = (t0 // 8) * 8
        wave = [row_entries(t) for t in range(t0, t0 + 8)]
        ntiles = (t0 + 8 + KB - 1) // KB
        for b in range(ntiles):
            k1 = (b + 1) * KB; k0 = b * KB
            units = []
            for runs in wave:
                n = sum(hi - lo for lo, hi in runs if k0 <= hi - 1 < k1)
                tot_e += n
                # batches of 8 entries, each batch = ceil(min(8, rem)/U) units
                u = 0; rem = n
                while rem > 0:
                    take = min(8, rem); u += -(-take // U); rem -= take
                units.append(u)
            tot_slots += max(units) * U * 8
    return tot_e / max(tot_slots, 1)
for KB in (128, 256, 512):
    print("KB", KB, "U=4 eff", round(sim(KB, 4), 3), " U=2 eff", round(sim(KB, 2), 3), " U=1 eff", round(sim(KB, 1), 3))

def simB(KB, G=8, rows=range(64, T, 97)):
    tot_e = 0; tot_slots = 0
    for t in rows:
        runs = row_entries(t)
        ntiles = (t + 1 + KB - 1) // KB
        for b in range(ntiles):
            k1 = (b + 1) * KB; k0 = b * KB
            n = sum(hi - lo for lo, hi in runs if k0 <= hi - 1 < k1)
            tot_e += n
            tot_slots += G * (-(-n // G))
    return tot_e / max(tot_slots, 1)
for KB in (128, 256, 512, 1024):
    print("mapping B (wave per row) KB", KB, "eff", round(simB(KB), 3))
