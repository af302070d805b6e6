import os, sys, subprocess
sys.path.insert(0, "end-to-end-slu_amd")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from slu_hip import ops
    H = 128
    sync_between = sys.argv[2] == "1"
    ns = int(sys.argv[3])
    for spec in sys.argv[4:]:
        T, B, D = [int(v) for v in spec.split(",")]
        torch.manual_seed(T * 11 + B)
        gx = torch.randn(T, B, D * 3 * H, device="cuda")
        wf, bf = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1
        wr, br = (torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1) if D == 2 else (None, None)
        print("   ", spec, "launch", flush=True)
        one, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, seq_tiles=1)
        if sync_between:
            torch.cuda.synchronize(); print("      one ok", flush=True)
        two, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, seq_tiles=2)
        torch.cuda.synchronize()
        print("      ok equal", torch.equal(one, two), flush=True)
    sys.exit(0)
cases = [("full sequence", "0", "1", ["75,128,2", "38,37,2", "1,70,2", "2,33,1", "301,2560,2"]),
         ("only big", "0", "1", ["301,2560,2"]),
         ("only big, sync between", "1", "1", ["301,2560,2"]),
         ("D=1 then big", "0", "1", ["2,33,1", "301,2560,2"]),
         ("T=1 then big", "0", "1", ["1,70,2", "301,2560,2"]),
         ("small then big", "0", "1", ["75,128,2", "301,2560,2"]),
         ("full sequence, sync between", "1", "1", ["75,128,2", "38,37,2", "1,70,2", "2,33,1", "301,2560,2"])]
for name, sync, ns, specs in cases:
    print("==", name, flush=True)
    r = subprocess.run([sys.executable, __file__, "--child", sync, ns] + specs, capture_output=True, text=True)
    print(r.stdout, end="")
    print("   rc", r.returncode, (r.stderr.strip().splitlines() or [""])[0][:200], flush=True)
