import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for l in d["config"]["coalescer"]["legs"]: print(l["caller_threads"], l["in_flight_per_thread"], l["queries_per_s"], l["mean_batch"], l["leader_ms_per_batch"])
