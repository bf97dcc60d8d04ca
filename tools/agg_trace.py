"""Aggregate gpurun_out/step_trace.txt by kernel: total us, launches, mean us (usage: python tools/agg_trace.py [file] [other file to diff])."""
import re, sys, collections
def agg(path):
    a = collections.OrderedDict()
    for l in open(path):
        p = l.split(None, 3)
        if len(p) < 4 or not p[0][0].isdigit():
            continue
        du = float(p[1]); name = re.sub(r'\(.*', '', p[3].rsplit(None, 1)[0].strip())
        e = a.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += du
    return a
A = agg(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_trace.txt")
B = agg(sys.argv[2]) if len(sys.argv) > 2 else None
tot = sum(v[1] for v in A.values())
for n, (c, t) in sorted(A.items(), key=lambda kv: -kv[1][1])[:48]:
    extra = ""
    if B is not None and n in B:
        extra = "   was %8.1f (%+.1f)" % (B[n][1], t - B[n][1])
    print("%8.1f us %4d x %7.1f  %-70s%s" % (t, c, t / c, n[:70], extra))
print("total %.1f us" % tot, ("(was %.1f)" % sum(v[1] for v in B.values())) if B else "")
