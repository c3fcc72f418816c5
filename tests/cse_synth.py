"""Deterministic small GTF / VCF / FASTA / BAM quartets for `cis-splice-effects identify` tests (TEST INFRASTRUCTURE).

The model has genes on both strands with alternative transcripts (shared and skipped exons), reads spliced across
annotated and novel junctions, and SNVs planted at/near splice sites, inside exons, deep in introns and between genes."""
import os
import random

import bamio

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def build(outdir, seed=1, n_genes=10, reads_per_junction=6, contigs=(("chrA", 120000), ("chrB", 90000))):
    rnd = random.Random(seed)
    os.makedirs(outdir, exist_ok=True)
    seqs = {name: [rnd.choice("ACGT") for _ in range(length)] for name, length in contigs}
    gtf_lines = ["#synthetic annotation seed %d" % seed]
    genes = []          # (contig, strand, [exons (start1,end1 inclusive, 1-based)], transcripts [[exon idx...]])
    for g in range(n_genes):
        cname, clen = contigs[g % len(contigs)]
        strand = "+-"[(g // len(contigs)) % 2]
        n_ex = rnd.randint(3, 7)
        pos = 2000 + (g // len(contigs)) * (clen // (n_genes // len(contigs) + 1)) + rnd.randint(0, 500)
        exons = []
        for _ in range(n_ex):
            ln = rnd.randint(80, 260)
            exons.append((pos, pos + ln - 1))
            pos += ln + rnd.randint(120, 2500)
        if exons[-1][1] + 500 >= clen:
            continue
        # transcripts: full, one with a skipped exon, one truncated, sometimes a single-exon one
        txs = [list(range(n_ex))]
        if n_ex >= 4:
            k = rnd.randint(1, n_ex - 2)
            txs.append([i for i in range(n_ex) if i != k])
        if n_ex >= 3:
            txs.append(list(range(rnd.randint(0, 1), n_ex - rnd.randint(0, 1))))
        if rnd.random() < 0.4:
            txs.append([rnd.randrange(n_ex)])
        genes.append((cname, strand, exons, txs))
        # canonical motifs on the genome: intron = (exon_end+1 .. next_start-1), 1-based
        s = seqs[cname]
        for a, b in zip(exons[:-1], exons[1:]):
            i0, i1 = a[1], b[0] - 2          # 0-based first / last-but-one base of the intron
            don, acc = ("GT", "AG") if strand == "+" else ("CT", "AC")
            s[i0], s[i0 + 1] = don[0], don[1]
            s[i1], s[i1 + 1] = acc[0], acc[1]
        for t, idxs in enumerate(txs):
            order = idxs if rnd.random() < 0.5 else list(reversed(idxs))     # exon lines need not be sorted in the file
            for i in order:
                attrs = 'gene_id "G%03d"; gene_name "GENE%d"; transcript_id "T%03d_%d"; exon_number "%d";' % (g, g, g, t, i + 1)
                if rnd.random() < 0.15:
                    attrs = 'transcript_id "T%03d_%d"; gene_name "GENE%d"; gene_id "G%03d"' % (g, t, g, g)
                gtf_lines.append("\t".join([cname, "synth", "exon", str(exons[i][0]), str(exons[i][1]), ".", strand, ".", attrs]))
            if t == 0:
                gtf_lines.append("\t".join([cname, "synth", "transcript", str(exons[0][0]), str(exons[-1][1]), ".", strand, ".", 'gene_id "G%03d"; transcript_id "T%03d_%d";' % (g, g, t)]))
    with open(os.path.join(outdir, "ann.gtf"), "w") as f:
        f.write("\n".join(gtf_lines) + "\n")
    # FASTA + .fai
    with open(os.path.join(outdir, "ref.fa"), "w") as f, open(os.path.join(outdir, "ref.fa.fai"), "w") as fi:
        off = 0
        for name, length in contigs:
            hdr = ">%s synthetic\n" % name
            f.write(hdr)
            off += len(hdr)
            fi.write("%s\t%d\t%d\t60\t61\n" % (name, length, off))
            s = "".join(seqs[name])
            for k in range(0, length, 60):
                f.write(s[k:k + 60] + "\n")
            off += length + (length + 59) // 60
    # reads
    tid_of = {name: i for i, (name, _) in enumerate(contigs)}
    recs = []
    q = 0
    for cname, strand, exons, txs in genes:
        pairs = set()
        for idxs in txs:
            for a, b in zip(idxs[:-1], idxs[1:]):
                pairs.add((a, b))
        for a in range(len(exons) - 2):          # novel exon-skipping combinations
            if rnd.random() < 0.5:
                pairs.add((a, a + 2))
        for a, b in sorted(pairs):
            don_end, acc_start = exons[a][1], exons[b][0]          # 1-based inclusive exon end / next exon start
            shift_d = rnd.choice([0, 0, 0, 3, -4])                   # some novel donors / acceptors
            shift_a = rnd.choice([0, 0, 0, 5, -2])
            jstart, jend = don_end + shift_d, acc_start - 1 + shift_a    # 0-based intron [jstart, jend)
            if jend - jstart < 60:
                continue
            for _ in range(rnd.randint(1, reads_per_junction)):
                la, lb = rnd.randint(5, 60), rnd.randint(5, 60)
                pos0 = jstart - la
                cigar = "%dM%dN%dM" % (la, jend - jstart, lb)
                if rnd.random() < 0.1:
                    cigar = "3S" + cigar
                flag = rnd.choice([99, 147, 83, 163, 0, 16])
                tag = rnd.random()
                aux = bamio.tagA("XS", strand) if tag < 0.8 else (bamio.tagA("XS", "+-"[strand == "+"]) if tag < 0.9 else b"")
                recs.append((tid_of[cname], pos0, bamio.record(tid_of[cname], pos0, cigar, flag=flag, qname="r%05d" % q, aux=aux)))
                q += 1
        for _ in range(20):                       # unspliced background
            e = rnd.choice(exons)
            pos0 = e[0] - 1 + rnd.randint(0, max(0, e[1] - e[0] - 50))
            recs.append((tid_of[cname], pos0, bamio.record(tid_of[cname], pos0, "50M", qname="u%05d" % q, aux=bamio.tagA("XS", strand))))
            q += 1
    recs.sort(key=lambda r: (r[0], r[1]))
    bam = os.path.join(outdir, "aln.bam")
    bamio.write_bam(bam, list(contigs), [r[2] for r in recs], block=5000)
    from regtools_amd import synth
    synth.index(bam)
    # variants (sorted by contig order then position)
    var = []
    for cname, strand, exons, txs in genes:
        for (s1, e1) in exons:
            for d in (-3, -2, -1, 0, 1, 2, 3, 6):
                if rnd.random() < 0.35:
                    var.append((cname, s1 + d))
                if rnd.random() < 0.35:
                    var.append((cname, e1 + d))
            if rnd.random() < 0.5:
                var.append((cname, (s1 + e1) // 2))
        for a, b in zip(exons[:-1], exons[1:]):
            if rnd.random() < 0.5:
                var.append((cname, (a[1] + b[0]) // 2))
    for cname, clen in contigs:
        for _ in range(6):
            var.append((cname, rnd.randint(1, clen - 1)))
    var = sorted(set(var), key=lambda v: (tid_of[v[0]], v[1]))
    with open(os.path.join(outdir, "var.vcf"), "w") as f:
        f.write("##fileformat=VCFv4.1\n")
        for name, length in contigs:
            f.write("##contig=<ID=%s,length=%d>\n" % (name, length))
        f.write('##INFO=<ID=DP,Number=1,Type=Integer,Description="Depth">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for k, (c, p) in enumerate(var):
            ref = seqs[c][p - 1]
            f.write("%s\t%d\t.\t%s\t%s\t50\tPASS\t%s\n" % (c, p, ref, COMP[ref], "DP=%d" % (10 + k % 7) if k % 3 else "."))
    return dict(vcf=os.path.join(outdir, "var.vcf"), bam=bam, fasta=os.path.join(outdir, "ref.fa"), gtf=os.path.join(outdir, "ann.gtf"))
