"""A SAM file with every kind of record the reference's reader tells apart (tests/test_gpu_sam.py, tests/test_sam_oracle.py)."""
import numpy as np


def random_sam(seed, n_lines, names, crlf=False):
    rng = np.random.default_rng(seed)
    out = [b"@HD\tVN:1.0\tSO:unsorted\n"] + [f"@SQ\tSN:{n}\tLN:100000\n".encode() for n in names[:5]]
    eol = b"\r\n" if crlf else b"\n"
    for i in range(n_lines):
        L = int(rng.integers(30, 260))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        qual = "I" * L
        flag = int(rng.choice([0, 16, 4, 256, 272, 99, 147, 1024]))
        name = names[int(rng.integers(0, len(names)))] if rng.random() < 0.9 else ("unknown_%d" % rng.integers(0, 50))
        pos = int(rng.integers(0, 90000)) if rng.random() < 0.98 else 0
        kind = rng.integers(0, 6)
        tags = ["XA:i:0"] if rng.random() < 0.5 else []
        cigar = f"{L}M"
        if kind == 1:                                  # mismatches
            k = int(rng.integers(1, 20))
            gaps = rng.multinomial(L - k, np.ones(k + 1) / (k + 1)) if L > k else None
            if gaps is not None:
                md = str(gaps[0]) + "".join("ACGT"[int(rng.integers(0, 4))] + str(int(x)) for x in gaps[1:])
                tags.append("MD:Z:" + md)
        elif kind == 2 and L > 40:                     # an insertion (or two)
            i1 = int(rng.integers(5, L - 30)); il = int(rng.integers(1, 4))
            cigar = f"{i1}M{il}I{L - i1 - il}M"
            tags.append(f"MD:Z:{L - il}")
        elif kind == 3 and L > 40:                     # a deletion and a mismatch right behind it
            d1 = int(rng.integers(5, L - 30)); dl = int(rng.integers(1, 18))
            cigar = f"{d1}M{dl}D{L - d1}M"
            tags.append("MD:Z:%d^%s0T%d" % (d1, "".join(rng.choice(list("ACGT"), size=dl)), L - d1 - 1))
        elif kind == 4:
            tags.append(f"MD:Z:{L}")
            tags.append("NM:i:0")
        line = "\t".join([f"read{i}/1", str(flag), name, str(pos), "255", cigar, "=", str(pos + 200), "300", seq, qual] + tags)
        out.append(line.encode() + eol)
    return np.frombuffer(b"".join(out), dtype=np.uint8)
