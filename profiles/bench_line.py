import json, sys
for f in sys.argv[1:]:
    j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, "ms/step %.1f" % j["ms_per_step"], "spmv %.4f ms frac %.3f" % (j["roofline"]["avg_ms"], j["roofline"]["frac"]), "relax %.4f ms frac %.3f" % (j["roofline_relax"]["avg_ms"], j["roofline_relax"]["frac"]), "setup %.1f" % j["setup_s"], "its", j["iterations_per_solve"])
