"""Synthetic fixtures for the k=5 scenario tests (the role of the reference's
tests/helpers.py, which needs pysam): a deterministic reference sequence and a
minimal BAM writer (BGZF blocks via zlib) so that the engine's own BAM reader is
exercised end to end."""
import hashlib
import struct
import zlib

_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def make_ref_fasta(path, chrom="chr1", length=200):
    """Non-repetitive deterministic sequence: base i = "ACGT"[md5(str(i)) mod 4]
    (same definition as the reference's synthetic tests, so scenarios carry over)."""
    seq = "".join("ACGT"[int(hashlib.md5(str(i).encode()).hexdigest(), 16) % 4] for i in range(length))
    with open(path, "w") as fh:
        fh.write(f">{chrom}\n{seq}\n")
    return seq


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25                      # 18 header + comp + 8 trailer - 1
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def write_bam(path, chroms, reads):
    """reads: dicts with name, seq and optional flag (0), ref (0; -1 unplaced),
    pos (0-based), mapq (60), cigar ([(0, len)]), qual (True: has qualities),
    aux (bytes: optional fields in BAM encoding, e.g. b"NMC\x01SAZchr1,5,+,10M,60,0;\0").
    Records are written in the given order (sort them yourself when needed)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in chroms)
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(chroms)))
    for n, l in chroms:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    for r in reads:
        seq = r["seq"]
        name = r["name"].encode() + b"\0"
        cigar = r.get("cigar", [(0, len(seq))] if seq else [])
        if r.get("flag", 0) & 4:
            cigar = []
        sq = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            sq[i >> 1] |= _NT16[ch.upper()] << (0 if i & 1 else 4)
        qual = (b"\x28" if r.get("qual", True) else b"\xff") * len(seq)
        body = struct.pack("<iiBBHHHiiii", r.get("ref", 0), r.get("pos", 0), len(name), r.get("mapq", 60), 4680,
                           len(cigar), r.get("flag", 0), len(seq), -1, -1, 0)
        body += name + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar) + bytes(sq) + qual + r.get("aux", b"")
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as fh:
        data = bytes(out)
        for i in range(0, len(data), 60000):
            fh.write(_bgzf_block(data[i:i + 60000]))
        fh.write(_bgzf_block(b""))               # EOF marker
