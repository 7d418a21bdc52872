"""Developer probe (GPU box): what a plain device-to-device copy and a read-only pass reach on this part -- the practical
HBM ceilings the rollout kernel's 6.1-6.3 TB/s is compared with."""
import json, torch
dev = torch.device("cuda:0")
out = {}
for mib in (512, 2048):
    n = mib * (1 << 20) // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    def timed(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t_copy = timed(lambda: b.copy_(a))
    t_read = timed(lambda: a.sum())
    t_fill = timed(lambda: b.fill_(1.0))
    out[f"{mib}_MiB"] = dict(copy_TB_per_s=2 * a.numel() * 4 / t_copy / 1e12, read_TB_per_s=a.numel() * 4 / t_read / 1e12,
                             write_TB_per_s=a.numel() * 4 / t_fill / 1e12)
print(json.dumps(out, indent=1))
