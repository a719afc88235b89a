"""Dev measurement (GPU box): is a same-box A/B of two trees decided by the kernels or by the clock the box grants them?
Alternates the trees (A B A B ...), each pass one child process looping LLFQVAE_V4.tokenize on a workload for a few seconds while
this process samples rocm-smi (shader clock, socket power, temperature).  Prints per pass: ms/launch, mean sclk, mean power.
   python scripts/dev/clock_ab.py <workload> <passes> <tree A> <tree B> ...      (a tree = a checkout root with a built library)
   python scripts/dev/clock_ab.py --child <tree> <workload> <seconds>            (internal)"""
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path


def child(tree, wl, seconds, opts=""):
    sys.path.insert(0, str(Path(tree).resolve()))
    import torch
    import lipvq_vae_amd  # noqa: F401
    for kv in filter(None, opts.split(",")):
        from lipvq_vae_amd import ops
        ops.set_option(*kv.split("="))
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    from bench import WORKLOADS, trained_like_
    B, T, A, D, K = WORKLOADS[wl]
    torch.manual_seed(0)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    trained_like_(model, A)
    x = torch.randn(B * T, A, device="cuda")
    for _ in range(300):
        model.tokenize(x)
    torch.cuda.synchronize()
    print("READY", flush=True)
    t_end, out = time.time() + seconds, []
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            model.tokenize(x)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 500)
    print("MS " + " ".join(f"{v:.4f}" for v in out), flush=True)


def smi_sample():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        c = d[sorted(d)[0]]
        sclk = power = temp = None
        for k, v in c.items():
            kl = k.lower()
            if kl.startswith("sclk clock speed"):
                sclk = float(str(v).strip("()Mhz "))
            elif "power" in kl and "(w)" in kl:
                power = float(v)
            elif "temperature" in kl and "hotspot" in kl or "junction" in kl:
                temp = float(v)
        return sclk, power, temp
    except Exception as ex:                                        # the measurement still reports times without rocm-smi
        return None, None, None


def main():
    wl, passes, trees = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True)
    print("# rocm-smi idle sample:", r.stdout.strip()[:1500], r.stderr.strip()[:300])
    print(f"{'pass':>4} {'tree':60s} {'ms/launch (median of windows)':>30} {'min':>8} {'sclk MHz':>9} {'power W':>8} {'temp C':>7}")
    for p in range(passes):
        for tree in trees:
            parts = tree.split("::")                                # "<tree>[::<library>[::opt=val,opt=val]]": that tree's host code on
            root, lib, opts = parts[0], (parts[1] if len(parts) > 1 else ""), (parts[2] if len(parts) > 2 else "")   # another build / options
            env = dict(os.environ)
            if lib:
                env["LIPVQ_HIP_LIBRARY"] = lib
            proc = subprocess.Popen([sys.executable, __file__, "--child", root, wl, "4", opts], stdout=subprocess.PIPE, text=True, env=env)
            samples, stop = [], threading.Event()
            line = proc.stdout.readline()
            assert line.startswith("READY"), line

            def sampler():
                while not stop.is_set():
                    samples.append(smi_sample())
                    time.sleep(0.2)
            th = threading.Thread(target=sampler)
            th.start()
            ms = [float(v) for v in proc.stdout.readline().split()[1:]]
            stop.set()
            th.join()
            proc.wait()
            ms.sort()

            def mean(i):
                v = [s[i] for s in samples if s[i] is not None]
                return sum(v) / len(v) if v else float("nan")
            print(f"{p:4d} {tree:60s} {ms[len(ms) // 2]:30.4f} {ms[0]:8.4f} {mean(0):9.0f} {mean(1):8.0f} {mean(2):7.0f}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "")
    else:
        main()
