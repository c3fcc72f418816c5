"""Small VCF inputs for what the reference SAYS while it reads a VCF and how it ends (tests/test_vcf_diagnostics.py; the expected lines come from the real
reference: tests/golden/make_golden_vcf_diag.py -> tests/golden/vcf_writer/diagnostics.json)."""

H = ('##fileformat=VCFv4.2\n##contig=<ID=1>\n##FILTER=<ID=q10,Description="q">\n##INFO=<ID=DP,Number=1,Type=Integer,Description="d">\n'
     '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">\n##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">\n')
COLS = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
ONE = COLS + "\tFORMAT\tS0\n"
TWO = COLS + "\tFORMAT\tS0\tS1\n"


def rec(pos, info="DP=1", fmt=None, samples=(), chrom="1", filt="."):
    cols = [chrom, str(pos), ".", "A", "G", ".", filt, info]
    if fmt is not None: cols += [fmt] + list(samples)
    return "\t".join(cols) + "\n"


CASES = {
    # names the header does not declare: one line per name and role, the first time only
    "undeclared_once": H + ONE + rec(10, "XX=1;DP=2", "GT:ZZ", ["0/1:7"], chrom="7", filt="lowq") + rec(20, "XX=3", "GT:ZZ", ["0/1:8"], chrom="7", filt="lowq;q10")
                       + rec(30, "YY", "GT", ["0/1"], chrom="8"),
    "filter_named_like_info": H + ONE + rec(10, "DP=1", "GT", ["0/1"], filt="DP") + rec(20, "q10=4", "GT", ["0/1"]),
    # the four keys of the annotated VCF are declared before the records are read
    "carries_annotation_keys": H + ONE + rec(10, "genes=G1;transcripts=T1;distances=5;annotations=x", "GT", ["0/1"]),
    # sample columns that do not fit: the read loop ends there
    "fewer_samples": H + TWO + rec(10, fmt="GT", samples=["0/1", "1/1"]) + rec(20, "NEW=1", "GT", ["0/1"]) + rec(30, fmt="GT", samples=["0/1", "1/1"]),
    "format_without_samples": H + ONE + rec(10, fmt="GT", samples=["0/1"]) + rec(20, "NEW=1", "GT") + rec(30, fmt="GT", samples=["0/1"]),
    "more_samples_than_header": H + ONE + rec(10, fmt="GT", samples=["0/1", "1/1"]) + rec(20, fmt="GT", samples=["0/1"]),
    # a record without FORMAT under a header with samples is read, and skipped by the writer
    "no_format_column": H + ONE + rec(10, "A1=1") + rec(20, "A2=1", "GT", ["0/1"]),
    # htslib ends the process itself
    "too_many_fields": H + ONE + rec(10, fmt="GT", samples=["0/1"]) + rec(20, "NEW=1", "GT", ["0/1:3"]) + rec(30, fmt="GT", samples=["0/1"]),
    "too_many_fields_in_a_short_record": H + TWO + rec(10, fmt="GT", samples=["0/1", "1/1"]) + rec(20, "NEW=1", "GT:DP", ["0/1:3:4"]),
    "flag_in_format": H + '##FORMAT=<ID=FL,Number=0,Type=Flag,Description="f">\n' + ONE + rec(10, fmt="GT", samples=["0/1"]) + rec(20, "NEW=1", "GT:FL", ["0/1:1"]),
    "format_without_type": H + '##FORMAT=<ID=NT,Number=1,Description="f">\n' + ONE + rec(10, "NEW=1", "GT:NT", ["0/1:1"]),
    "too_many_fields_before_flag": H + '##FORMAT=<ID=FL,Number=0,Type=Flag,Description="f">\n' + ONE + rec(10, "NEW=1", "GT:FL", ["0/1:1:2"]),
    # the header
    "unknown_type": H + '##INFO=<ID=W1,Number=1,Type=Whatever,Description="w">\n##FORMAT=<ID=W2,Number=1,Type="Integer",Description="w">\n' + ONE
                    + rec(10, "W1=a", "GT:W2", ["0/1:5"]),
    "character_type": H + '##INFO=<ID=C1,Number=1,Type=Character,Description="c">\n' + ONE + rec(10, "C1=a", "GT", ["0/1"]),
    "pl_not_per_genotype": H + '##FORMAT=<ID=PL,Number=3,Type=Integer,Description="p">\n' + ONE + rec(10, fmt="GT:PL", samples=["0/1:1,2,3"]),
    "pl_per_genotype": H + '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="p">\n' + ONE + rec(10, fmt="GT:PL", samples=["0/1:1,2,3"]),
    "gl_is_not_looked_at": H + '##FORMAT=<ID=GL,Number=3,Type=Float,Description="p">\n' + ONE + rec(10, fmt="GT:GL", samples=["0/1:1,2,3"]),
    "idx_not_a_number": H + '##INFO=<ID=I1,Number=1,Type=Integer,Description="i",IDX=x7>\n' + ONE + rec(10, "I1=4", "GT", ["0/1"]),
    # (a ##contig line with such an IDX leaves its name in upstream's dictionary without a number: whatever is printed for a contig afterwards is undefined)
    "conflicting_idx": H + '##INFO=<ID=I1,Number=1,Type=Integer,Description="i",IDX=1>\n' + ONE + rec(10, fmt="GT", samples=["0/1"]),
    "duplicated_sample": H + COLS + "\tFORMAT\tS0\tS0\n" + rec(10, fmt="GT", samples=["0/1", "0/1"]),
    "empty_sample_name": H + COLS + "\tFORMAT\tS0\t\n" + rec(10, fmt="GT", samples=["0/1"]),
    "no_sample_line": H + rec(10),
    "ends_inside_the_header": H,
    "line_that_does_not_scan": H + '##weird=<a=b>   \n##INFO=<ID=Z9,Number=1,Type=Integer,Description="z">\n' + ONE + rec(10, "Z9=1", "GT", ["0/1"]),
    "first_line_not_fileformat": '##fileformat=VCFv4.2\n'.replace("fileformat=VCFv4.2", "fileformat=VCF<") + H.split("\n", 1)[1] + ONE + rec(10, fmt="GT", samples=["0/1"]),
    # numbers written past their slot land on the tape behind it (an Integer GT has one place per sample and "0/2" holds two numbers)
    "integer_gt_runs_into_the_next_key": H.replace('ID=GT,Number=1,Type=String', 'ID=GT,Number=1,Type=Integer') + '##FORMAT=<ID=FT,Number=1,Type=String,Description="f">\n' + TWO
                                         + rec(10, fmt="GT:FT", samples=["2/10:a", "0/2"]),
}
