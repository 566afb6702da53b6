"""What one halo exchange costs on the device: two (and four) PROCESSES on GPU 0, mailboxes shared through hipIpc
(tests/two_process_rank.py, mode compare) -> gpurun_out/p2p_exchange.json (copied to profiles/<round>_p2p_exchange.json;
bench.py quotes it in rank_proxy.p2p_exchange).  Never touches the GPU itself: the ranks are child processes."""
import json, os, sys
sys.path.insert(0, "tests")
import rank_launcher
out = {"what": "padne_ctx_halo_exchange_time: average device time of one exchange over 300 queued back to back, processes on ONE "
               "GPU (stores through the device's memory, not xGMI); the all-gather figure goes through gloo and the host"}
for world in (2, 4):
    ans = rank_launcher.run({"script": "tests/two_process_rank.py", "args": ["compare"], "n": world, "env": {}, "timeout": 400})
    res = []
    for o in ans["out"]:
        lines = [ln for ln in o.splitlines() if ln.startswith("RESULT ")]
        res.append(json.loads(lines[-1][7:]) if lines else None)
    print(world, ans["rc"], ans["seconds"], res, flush=True)
    if all(r is not None for r in res):
        out[f"world_{world}"] = {"p2p_exchange_us": [r["p2p_exchange_us"] for r in res],
                                 "allgather_exchange_us_gloo": [r["allgather_exchange_us_gloo"] for r in res],
                                 "exchange_slots_per_rank": res[0]["exchange_slots_per_rank"], "p2p": [r["p2p"] for r in res],
                                 "iterations": res[0]["iterations"]}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/p2p_exchange.json", "w"), indent=1)
