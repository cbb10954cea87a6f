import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hopmi
import torch.cuda.tunable as tunable
ok = hopmi.use_tuned_gemms()
print("use_tuned_gemms:", ok, "enabled", tunable.is_enabled(), "tuning", tunable.tuning_is_enabled(), "file", tunable.get_filename())
print("validators now:", tunable.get_validators())
a = torch.randn(4352, 992, device="cuda"); w = torch.randn(2100, 992, device="cuda"); b = torch.randn(2100, device="cuda")
y = torch.nn.functional.linear(a, w, b)
torch.cuda.synchronize()
res = tunable.get_results()
print("results loaded:", len(res))
print([r for r in res if "2100_4352_992" in r[1]][:3])
