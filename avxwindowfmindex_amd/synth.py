"""Seeded synthetic texts and query sets, numpy side.

splitmix64 is counter based: the i-th output of a stream seeded with ``s`` is
``mix(s + (i+1)*GOLDEN)``, so every character can be generated independently
(the HIP generators in csrc/awfm_synth.hip compute exactly the same values).

  text char i          = alphabet[ mix(seed + (i+1)*G) % |A| ]
  query j stream state = q_j = mix(seed_q + j)            ("hashed once")
  random query j       : char c = alphabet[ mix(q_j + (c+1)*G) % |A| ]
  planted query j      : offset = mix(q_j + G) % (n-L+1), chars = text[offset:offset+L]
  mixed length j       : L = lo + mix(q_j + 2^63 + G) % (hi-lo+1)
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
DNA_ALPHABET = b"acgt"
AMINO_ALPHABET = b"acdefghiklmnpqrstvwy"


def mix64(z):
    z = np.asarray(z, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def stream(seed, start, count):
    """outputs start..start+count-1 of the splitmix64 stream seeded with ``seed``"""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        return mix64(np.uint64(seed) + idx * GOLDEN)


def text(seed, n, alphabet=DNA_ALPHABET, start=0):
    lut = np.frombuffer(alphabet, dtype=np.uint8)
    return lut[(stream(seed, start, n) % np.uint64(len(alphabet))).astype(np.int64)]


GENOME_BLOCK, GENOME_RUNS = 1024, 24
_SALT_BLOCK, _SALT_FAM_A, _SALT_FAM_B, _SALT_DIV = 0xB10C5A17, 0xFA111A5, 0xFA111B5, 0xD17E26E5
_SALT_RUN_START, _SALT_RUN_LEN = 0x52554E53, 0x52554E4C


def genome_runs(seed, n):
    """(starts, lengths) of the 24 runs of 'n' of genome_text(seed, n)"""
    longest = max(1, min(n // 64, 10_000_000))
    shortest = max(1, min(longest // 100, 100_000))
    r = np.arange(GENOME_RUNS, dtype=np.uint64)
    with np.errstate(over="ignore"):
        starts = mix64(np.uint64(seed) + np.uint64(_SALT_RUN_START) + r) % np.uint64(n)
        lengths = np.uint64(shortest) + mix64(np.uint64(seed) + np.uint64(_SALT_RUN_LEN) + r) % np.uint64(longest - shortest + 1)
    return starts, lengths


def genome_text(seed, n):
    """genome-shaped nucleotide text (csrc/awfm_synth.hip synthGenomeTextKernel computes the same characters): blocks of
    1024 characters that are, by their hash, a window into a 300-character family consensus repeated end to end (10 % of
    the blocks, 10 % divergence), a window of a 6000-character consensus (15 %, 5 %), a tandem repeat of a unit of 2..64
    characters (3 %, 2 %) or unique sequence (72 %: the characters of text(seed, n)); 24 runs of 'n' on top"""
    lut = np.frombuffer(DNA_ALPHABET, dtype=np.uint8)
    i = np.arange(n, dtype=np.uint64)
    b, j = i // np.uint64(GENOME_BLOCK), i % np.uint64(GENOME_BLOCK)
    with np.errstate(over="ignore"):
        s = np.uint64(seed)
        hb = mix64(s + np.uint64(_SALT_BLOCK) + (b + np.uint64(1)) * GOLDEN)
        kind = hb % np.uint64(100)
        pick = hb >> np.uint64(8)
        fam_a = mix64(s + np.uint64(_SALT_FAM_A) + ((pick + j) % np.uint64(300) + np.uint64(1)) * GOLDEN) % np.uint64(4)
        fam_b = mix64(s + np.uint64(_SALT_FAM_B) + ((pick + j) % np.uint64(6000) + np.uint64(1)) * GOLDEN) % np.uint64(4)
        unit = np.uint64(2) + pick % np.uint64(63)
        tandem = mix64(hb + (j % unit + np.uint64(1)) * GOLDEN) % np.uint64(4)
        unique = mix64(s + (i + np.uint64(1)) * GOLDEN) % np.uint64(4)
        r = mix64(s + np.uint64(_SALT_DIV) + (i + np.uint64(1)) * GOLDEN)
    base = np.where(kind < 10, fam_a, np.where(kind < 25, fam_b, tandem))
    permille = np.where(kind < 10, 100, np.where(kind < 25, 50, 20)).astype(np.uint64)
    base = np.where(r % np.uint64(1000) < permille, (r >> np.uint64(16)) % np.uint64(4), base)
    out = lut[np.where(kind < 28, base, unique).astype(np.int64)]
    starts, lengths = genome_runs(seed, n)
    for st, ln in zip(starts.tolist(), lengths.tolist()):
        out[st:min(n, st + ln)] = ord("n")
    return out


def planted_queries_clean(seed_q, count, length, txt, first=0):
    """planted_queries with every character that is not a,c,g,t replaced by a seeded random letter"""
    q = planted_queries(seed_q, count, length, txt, first).copy()
    lut = np.frombuffer(DNA_ALPHABET, dtype=np.uint8)
    state = _qstate(seed_q, first, count)[:, None]
    with np.errstate(over="ignore"):
        z = mix64(state + (np.arange(2, length + 2, dtype=np.uint64) * GOLDEN)[None, :])
    plain = np.isin(q, lut)
    q[~plain] = lut[(z % np.uint64(4)).astype(np.int64)][~plain]
    return q


def planted_unique_offsets(seed_q, count, length, txt, text_seed, first=0):
    """the offsets awfmGpuSynthPlantedQueriesUnique takes its k-mers from: the first of up to 64 seeded offsets whose window lies
    in unique blocks of the genome-shaped text `txt` = genome_text(text_seed, n) and holds only a,c,g,t"""
    n = len(txt)
    q = _qstate(seed_q, first, count)
    lut = np.frombuffer(DNA_ALPHABET, dtype=np.uint8)
    plain = np.isin(txt, lut)
    bad_before = np.concatenate([[0], np.cumsum(~plain)])  # characters that are no a,c,g,t before position i
    out = np.zeros(count, dtype=np.uint64)
    done = np.zeros(count, dtype=bool)
    with np.errstate(over="ignore"):
        for t in range(64):
            off = mix64(q + np.uint64(t + 1) * GOLDEN) % np.uint64(n - length + 1)
            b0, b1 = off // np.uint64(GENOME_BLOCK), (off + np.uint64(length - 1)) // np.uint64(GENOME_BLOCK)
            k0 = mix64(np.uint64(text_seed) + np.uint64(_SALT_BLOCK) + (b0 + np.uint64(1)) * GOLDEN) % np.uint64(100)
            k1 = mix64(np.uint64(text_seed) + np.uint64(_SALT_BLOCK) + (b1 + np.uint64(1)) * GOLDEN) % np.uint64(100)
            o = off.astype(np.int64)
            ok = (k0 >= 28) & (k1 >= 28) & (bad_before[o + length] == bad_before[o])
            take = ~done & (ok | (t == 63))
            out[take] = off[take]
            done |= take
    return out


def _qstate(seed_q, first, count):
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed_q) + np.arange(first, first + count, dtype=np.uint64))


def random_queries(seed_q, count, length, alphabet=DNA_ALPHABET, first=0):
    """uint8[count, length] of uniform random k-mers (query ids first..first+count-1)"""
    lut = np.frombuffer(alphabet, dtype=np.uint8)
    q = _qstate(seed_q, first, count)[:, None]
    with np.errstate(over="ignore"):
        c = (np.arange(1, length + 1, dtype=np.uint64) * GOLDEN)[None, :]
        z = mix64(q + c)
    return lut[(z % np.uint64(len(alphabet))).astype(np.int64)]


def planted_offsets(seed_q, count, length, n, first=0):
    q = _qstate(seed_q, first, count)
    with np.errstate(over="ignore"):
        return mix64(q + GOLDEN) % np.uint64(n - length + 1)


def planted_queries(seed_q, count, length, txt, first=0):
    off = planted_offsets(seed_q, count, length, len(txt), first).astype(np.int64)
    return txt[off[:, None] + np.arange(length, dtype=np.int64)[None, :]]


def mixed_lengths(seed_q, count, lo=8, hi=30, first=0):
    q = _qstate(seed_q, first, count)
    with np.errstate(over="ignore"):
        return (np.uint64(lo) + mix64(q + np.uint64(1 << 63) + GOLDEN) % np.uint64(hi - lo + 1)).astype(np.int64)


def mixed_queries(seed_q, count, txt, alphabet=DNA_ALPHABET, lo=8, hi=30, first=0):
    """CSR (chars uint8, offsets uint64): even ids random, odd ids planted, lengths lo..hi"""
    lens = mixed_lengths(seed_q, count, lo, hi, first)
    offsets = np.zeros(count + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    chars = np.empty(int(offsets[-1]), dtype=np.uint8)
    rnd = random_queries(seed_q, count, hi, alphabet, first)
    n = len(txt)
    q = _qstate(seed_q, first, count)
    with np.errstate(over="ignore"):
        draw = mix64(q + GOLDEN)
    for j in range(count):
        L = int(lens[j])
        o = int(offsets[j])
        if (first + j) % 2 == 0:
            chars[o:o + L] = rnd[j, :L]
        else:
            s = int(draw[j] % np.uint64(n - L + 1))
            chars[o:o + L] = txt[s:s + L]
    return chars, offsets


def fixed_csr(q2d):
    """uint8[count, L] -> CSR (chars, offsets)"""
    count, L = q2d.shape
    return np.ascontiguousarray(q2d).reshape(-1), (np.arange(count + 1, dtype=np.uint64) * np.uint64(L))


def fnv1a(arr, h=0xCBF29CE484222325):
    """FNV-1a-64 of the array's bytes (slow pure-python; use for small arrays)"""
    for b in np.ascontiguousarray(arr).view(np.uint8).reshape(-1).tolist():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
