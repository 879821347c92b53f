import json, sys
b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = b["roofline"]["all_kernels"]
print(f"{b['dtype']:14s} graph {b['ms_per_step']:.3f} ms  eager {b['eager_ms_per_step']:.3f}  " + "  ".join(f"{x} {k[x]['ms_per_step']:.3f}" for x in ("fwd", "bwd_dgrad", "bwd_wgrad", "loss", "bwd_reduce")) + f"  {b['config']['backward']}  loss {b['final_loss']:.6e}")
