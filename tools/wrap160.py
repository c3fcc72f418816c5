#!/usr/bin/env python3
"""tools/wrap160.py FILE...: re-break source lines longer than 160 columns (C++: white space between tokens means nothing).  A trailing // comment moves to
its own line above; code is cut after the last ", " / "; " / " && " / " || " / " ? " / " : " / " = " that lies outside string and character literals and
in front of column 158, the rest continues one level deeper.  Preprocessor lines, lines ending in a backslash and lines it finds no safe cut in are
left alone (and reported)."""
import sys

LIMIT = 160


def safe_cuts(s):
    """positions just behind a separator, outside literals and comments"""
    out, i, n, q = [], 0, len(s), None
    while i < n:
        c = s[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'":
            q = c
        elif s.startswith("//", i) or s.startswith("/*", i):
            break
        else:
            for sep in (", ", "; ", " && ", " || ", " ? ", " : "):
                if s.startswith(sep, i):
                    out.append(i + len(sep))
        i += 1
    return out, q is not None


def comment_start(s):
    i, n, q = 0, len(s), None
    while i < n:
        c = s[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c
        elif s.startswith("//", i): return i
        i += 1
    return -1


def wrap(line, report):
    if len(line) <= LIMIT: return [line]
    st = line.lstrip()
    ind = line[: len(line) - len(st)]
    if st.startswith("#") or line.rstrip().endswith("\\"): report.append(line); return [line]
    if st.startswith("//"):
        words, out, cur = st[2:].split(" "), [], ind + "//"
        for w in words:
            if len(cur) + 1 + len(w) > LIMIT and cur.strip() != "//": out.append(cur); cur = ind + "// " + w
            else: cur = cur + " " + w if cur != ind + "//" or w else cur + " " + w
        out.append(cur)
        return [o.replace("//  ", "// ", 1) if o.startswith(ind + "//  ") and not st.startswith("//  ") else o for o in out]
    k = comment_start(line)
    pre = []
    if k > 0:
        code, com = line[:k].rstrip(), line[k:]
        if code.strip():
            pre = wrap(ind + com, report)
            line = code
            if len(line) <= LIMIT: return pre + [line]
    out = []
    cur = line
    first = True
    while len(cur) > LIMIT:
        cuts, open_q = safe_cuts(cur)
        cuts = [c for c in cuts if len(ind) + 8 < c <= LIMIT - 2]
        if not cuts: report.append(cur); break
        c = cuts[-1]
        out.append(cur[:c].rstrip())
        cur = ind + "    " + ("" if first else "") + cur[c:].lstrip()
        first = False
    out.append(cur)
    return pre + out


for path in sys.argv[1:]:
    report, res = [], []
    for l in open(path).read().split("\n"):
        res.extend(wrap(l, report))
    open(path, "w").write("\n".join(res))
    print(path, "left alone:", len(report))
    for r in report[:5]: print("   ", r[:140])
