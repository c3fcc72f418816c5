"""`cis-splice-effects identify -s XS` on a BAM in which ONE spliced read carries an aux field of unknown type in front of its strand tag (TEST INFRASTRUCTURE).
Upstream extracts every splice-relevant variant's window on its own: the first window -- in the order of the variants -- whose iterator reads that read
abort()s inside bam_aux_get (sam.c:1233-1252), behind the variant's echo; a read no window reads ends nothing."""
import hashlib
import os
import struct

import bamio
import cse_synth

ODD = b"ZZq\x01\x02\x03"          # tag ZZ, type 'q': no such type


def _fields(r):
    tid, pos = struct.unpack_from("<ii", r, 4)
    lq = r[12]
    nc = struct.unpack_from("<I", r, 16)[0] & 0xffff
    l_seq = struct.unpack_from("<i", r, 20)[0]
    cig = struct.unpack_from("<%dI" % nc, r, 36 + lq)
    end = pos + sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
    return tid, pos, end, cig, 36 + lq + 4 * nc + (l_seq + 1) // 2 + l_seq


def _inject(r):
    aux_off = _fields(r)[4]
    body = r[4:aux_off] + ODD + r[aux_off:]
    return struct.pack("<i", len(body)) + body


def build(td):
    """-> (quartet paths, {name: bam path}, digest of everything made).  Names: first / mid / last = the first, middle and last spliced read of the file;
    clear = a spliced read no variant's +-20 window reads (run with -w 20)."""
    from regtools_amd import synth
    q = cse_synth.build(os.path.join(td, "q"), seed=3, n_genes=8)
    contigs, recs = bamio.split_records(bamio.inflate_all(q["bam"]))
    names = [c[0] for c in contigs]
    spliced = [i for i, r in enumerate(recs) if any((c & 15) == 3 for c in _fields(r)[3])]
    var = [(l.split("\t")[0], int(l.split("\t")[1])) for l in open(q["vcf"]) if l[0] != "#"]
    picks = {"first": spliced[0], "mid": spliced[len(spliced) // 2], "last": spliced[-1]}
    # clear: one more spliced read with such a field, where no variant's +-20 window (-w 20) comes near it -- nothing happens
    clear_rec = None
    for tid, (cname, clen) in enumerate(contigs):
        for p0 in range(1000, clen - 2000, 250):
            if all(not (c == cname and abs(v - p0) < 800) for c, v in var):
                clear_rec = (tid, p0, bamio.record(tid, p0, "30M200N30M", qname="clear", aux=ODD + bamio.tagA("XS", "+")))
                break
        if clear_rec:
            break
    assert clear_rec, "no stretch without variants: another seed"
    picks["clear"] = None
    h = hashlib.sha256()
    out = {}
    for name, k in sorted(picks.items()):
        rr = list(recs)
        if k is None:
            keys = [_fields(r)[:2] for r in rr]
            at = sum(1 for t, q0 in keys if (t, q0) <= clear_rec[:2])
            rr.insert(at, clear_rec[2])
        else:
            rr[k] = _inject(rr[k])
        p = os.path.join(td, name + ".bam")
        h.update(bamio.write_bam(p, contigs, rr, block=5000))
        synth.index(p)
        out[name] = p
    for f in ("vcf", "gtf", "fasta"):
        h.update(open(q[f], "rb").read())
    return q, out, h.hexdigest()


def cases(td):
    """[(argv, output files)]"""
    q, bams, digest = build(td)
    out = []
    for name in sorted(bams):
        for extra in ([], ["-w", "20"]):
            tag = name + ("_w20" if extra else "")
            files = [os.path.join(td, tag + ".tsv"), os.path.join(td, tag + ".bed")]
            out.append((["cis-splice-effects", "identify", "-s", "XS"] + extra + ["-o", files[0], "-j", files[1], q["vcf"], bams[name], q["fasta"], q["gtf"]], files))
    return out, digest
