"""Deterministic synthetic genome + long reads as FASTA (for end-to-end runs of the reference host; SURVEY 8c Probe D used
the same kind of set: a few Mbp random genome with repeats, 5-80 kb reads, ~8 % error)."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = {65: 84, 67: 71, 71: 67, 84: 65}


def make_genome(rng, n_chr=3, chr_len=1_000_000, n_rep_families=6, rep_len=3000, copies=40, tandem=3):
    chrs = []
    fams = [BASES[rng.integers(0, 4, rep_len)] for _ in range(n_rep_families)]
    for _ in range(n_chr):
        g = BASES[rng.integers(0, 4, chr_len)].copy()
        for fam in fams:                                  # interspersed repeats, ~2 % diverged copies
            for _ in range(copies // n_chr):
                pos = int(rng.integers(0, chr_len - rep_len))
                cp = fam.copy()
                mut = rng.random(rep_len) < 0.02
                cp[mut] = BASES[rng.integers(0, 4, int(mut.sum()))]
                g[pos:pos + rep_len] = cp
        for _ in range(tandem):                           # tandem arrays: many anchors inside one max_dist_x window
            unit = BASES[rng.integers(0, 4, int(rng.integers(30, 200)))]
            n_units = int(rng.integers(50, 400))
            arr = np.tile(unit, n_units)
            pos = int(rng.integers(0, chr_len - len(arr)))
            g[pos:pos + len(arr)] = arr
        chrs.append(g)
    return chrs


def mutate(rng, seq, err):
    """Substitutions, insertions, deletions at total rate err (ONT-like mix 40/30/30)."""
    out = []
    r = rng.random(len(seq))
    ins_base = BASES[rng.integers(0, 4, len(seq))]
    sub_base = BASES[rng.integers(0, 4, len(seq))]
    keep = r >= err * 0.3                                  # deletions
    sub = (r >= err * 0.3) & (r < err * 0.7)
    ins = (r >= err * 0.7) & (r < err)
    s = seq.copy()
    s[sub] = sub_base[sub]
    # build with insertions: interleave
    idx = np.flatnonzero(keep)
    base = s[idx]
    add = ins[idx]
    total = len(base) + int(add.sum())
    res = np.empty(total, dtype=np.uint8)
    pos = np.arange(len(base)) + np.cumsum(add) - add
    res[pos] = base
    res[pos[add] + 1] = ins_base[idx][add]
    return res


def revcomp(seq):
    lut = np.zeros(256, dtype=np.uint8)
    for k, v in COMP.items():
        lut[k] = v
    return lut[seq[::-1]]


def write_fasta(path, records, width=0):
    with open(path, "wb") as fh:
        for name, seq in records:
            fh.write(b">" + name.encode() + b"\n")
            fh.write(seq.tobytes() + b"\n")


def simulate(ref_path, reads_path, seed=11, n_reads=400, len_lo=5_000, len_hi=80_000, err=0.08, **genome_kw):
    rng = np.random.default_rng(seed)
    chrs = make_genome(rng, **genome_kw)
    write_fasta(ref_path, [(f"chr{k + 1}", c) for k, c in enumerate(chrs)])
    recs = []
    total = 0
    for r in range(n_reads):
        c = int(rng.integers(0, len(chrs)))
        L = int(rng.integers(len_lo, len_hi))
        L = min(L, len(chrs[c]) - 1)
        st = int(rng.integers(0, len(chrs[c]) - L))
        seq = mutate(rng, chrs[c][st:st + L], err)
        if rng.random() < 0.5:
            seq = revcomp(seq)
        recs.append((f"read{r}_chr{c + 1}_{st}_{L}", seq))
        total += len(seq)
    write_fasta(reads_path, recs)
    return total
