import json, subprocess, sys
sys.path.insert(0, "tests")
import rank_launcher
ans = rank_launcher.run({"script": "tests/two_process_rank.py", "args": ["compare"], "n": 2, "env": {}, "timeout": 300})
print(ans["rc"], ans["seconds"])
for o in ans["out"]:
    print(o[-1500:])
ans = rank_launcher.run({"script": "tests/two_process_rank.py", "args": ["die"], "n": 2, "env": {"PADNE_P2P_TIMEOUT_MS": 3000}, "timeout": 100})
print(ans["rc"], ans["seconds"])
for o in ans["out"]:
    print(o[-1500:])
