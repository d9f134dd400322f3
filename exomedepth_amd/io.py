"""On-disk count-matrix container: what the path starts from.

The reference's read counting (getBamCounts, R/countBamInGranges.R:298-370) ends in an R data.frame: exon
coordinates plus one count column per BAM file.  Counting reads is out of scope here, but feeding a GPU from
R objects would starve it, so this module fixes a flat binary layout that can be memory-mapped and copied to
the device as is -- the count block is exactly the int32 [n_exons][n_samples] sample-minor matrix the batched
interface takes (ed_batch_fit / ed_batch_run), 4096-byte aligned.

    offset 0   magic  b"EDCOUNT1"
           8   uint32 version (1), uint32 n_chrom, uint64 n_exons, uint64 n_samples
          32   uint64 offsets of: chrom_off, start, end, chrom_names, sample_names, exon_names (0 = absent), counts
    then the sections: int32 chrom_off[n_chrom+1], int32 start[n_exons], int32 end[n_exons], string tables
    (uint32 count, then uint32 length + utf-8 bytes each), int32 counts[n_exons][n_samples].
Exons are stored in CallCNVs order (chromosome level, then mid-point; R/class_definition.R:323-336), so a file
maps 1:1 onto a Plan.
"""
import struct

import numpy as np

MAGIC = b"EDCOUNT1"
ALIGN = 4096


def _strtab(names):
    out = [struct.pack("<I", len(names))]
    for n in names:
        b = str(n).encode("utf-8")
        out.append(struct.pack("<I", len(b)))
        out.append(b)
    return b"".join(out)


def _read_strtab(buf, off):
    (n,) = struct.unpack_from("<I", buf, off)
    off += 4
    names = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4
        names.append(bytes(buf[off:off + ln]).decode("utf-8"))
        off += ln
    return names


def write_counts(path, chromosome, start, end, counts, sample_names=None, exon_names=None):
    """Order the exons as CallCNVs does and write the container.  counts: (n_exons, n_samples), integer valued."""
    from .api import chromosome_order
    counts = np.asarray(counts)
    if counts.ndim != 2 or counts.shape[0] != len(chromosome):
        raise ValueError("counts must be (n_exons, n_samples) with one row per exon")
    if np.any(counts != np.trunc(counts)) or counts.min(initial=0) < 0 or counts.max(initial=0) > 2**31 - 1:
        raise ValueError("counts must be non-negative integers below 2^31")
    order, levels, codes, chrom_off = chromosome_order(chromosome, start, end)
    E, S = counts.shape
    sample_names = list(sample_names) if sample_names is not None else ["sample%d" % (i + 1) for i in range(S)]
    if len(sample_names) != S:
        raise ValueError("one name per sample column")
    sections = [np.ascontiguousarray(chrom_off, dtype="<i4").tobytes(),
                np.ascontiguousarray(np.asarray(start)[order], dtype="<i4").tobytes(),
                np.ascontiguousarray(np.asarray(end)[order], dtype="<i4").tobytes(),
                _strtab(levels), _strtab(sample_names),
                _strtab(np.asarray(exon_names, dtype=object)[order]) if exon_names is not None else b""]
    offs = []
    pos = 32 + 8 * 7
    for sec in sections:
        offs.append(pos if sec else 0)
        pos += len(sec)
    counts_off = (pos + ALIGN - 1) // ALIGN * ALIGN
    offs.append(counts_off)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<IIQQ", 1, len(levels), E, S))
        f.write(struct.pack("<7Q", *offs))
        for sec in sections:
            f.write(sec)
        f.write(b"\0" * (counts_off - pos))
        np.ascontiguousarray(counts[order], dtype="<i4").tofile(f)
    return order


def read_counts(path, mmap=True):
    """Returns dict(chrom_off, start, end, chrom_names, sample_names, exon_names, counts); counts is a read-only
    memory map (or an array) of shape (n_exons, n_samples), int32, ready for Batch.fit / Batch.run."""
    with open(path, "rb") as f:
        head = f.read(32 + 56)
    if head[:8] != MAGIC:
        raise ValueError("%s is not an EDCOUNT1 file" % path)
    version, n_chrom, E, S = struct.unpack_from("<IIQQ", head, 8)
    if version != 1:
        raise ValueError("unsupported EDCOUNT version %d" % version)
    o_off, o_start, o_end, o_chr, o_smp, o_exn, o_cnt = struct.unpack_from("<7Q", head, 32)
    meta = np.memmap(path, dtype=np.uint8, mode="r", shape=(o_cnt,)) if o_cnt else np.zeros(0, np.uint8)
    out = {"chrom_off": np.frombuffer(meta, dtype="<i4", count=n_chrom + 1, offset=o_off).copy(),
           "start": np.frombuffer(meta, dtype="<i4", count=E, offset=o_start).copy(),
           "end": np.frombuffer(meta, dtype="<i4", count=E, offset=o_end).copy(),
           "chrom_names": _read_strtab(meta, o_chr), "sample_names": _read_strtab(meta, o_smp),
           "exon_names": _read_strtab(meta, o_exn) if o_exn else None}
    if mmap:
        out["counts"] = np.memmap(path, dtype="<i4", mode="r", offset=o_cnt, shape=(E, S))
    else:
        out["counts"] = np.fromfile(path, dtype="<i4", offset=o_cnt, count=E * S).reshape(E, S)
    return out
