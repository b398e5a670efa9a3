"""Print the self-validation fields of a bench line (file given as argv[1])."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "n_gpus", d["n_gpus"], "rccl_ranks", d.get("rccl_ranks"),
      "exchange_chosen", d.get("exchange_chosen"), "exchange_ab_ms", d.get("exchange_ab_ms"),
      "sharded_grad_check", d.get("sharded_grad_check"))
keep = ("max_err", "ok", "agrees", "gb_per_s", "ms", "overlap_in_use", "seconds")
for k, v in d.get("selfcheck", {}).items():
    print(" ", k, {kk: vv for kk, vv in v.items() if kk in keep} if isinstance(v, dict) else v)
