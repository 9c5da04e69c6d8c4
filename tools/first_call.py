#!/usr/bin/env python3
"""First-call latency of the model: reference-layout checkpoint (.pt: construct 711 buffers, strict load, move to the GPU,
fold / scale / cast at first use) vs packed-operand file (tools/pack_checkpoint.py --fold: file -> device).
Each variant runs in a fresh process (cold library, cold allocator); times are wall clock.
usage: first_call.py [--precision mixed]            (driver: writes the two files under /tmp, runs both variants)"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(cfg, ckpt):
    t0 = time.perf_counter()
    import torch
    from audiocodec.model import AudioCodec
    from bench import bench_inputs
    torch.cuda.init(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
    t1 = time.perf_counter()
    m = AudioCodec.load_from_checkpoint(cfg, ckpt).to("cuda:0").eval()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wavs = [w.cuda() for w in bench_inputs(2, 32000)]
    out = m.decode(m.encode(wavs)["codes_list"])["syn_wav_list"]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    out2 = m.decode(m.encode(wavs)["codes_list"])["syn_wav_list"]
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(json.dumps({"import+gpu_init_s": round(t1 - t0, 3), "load_s": round(t2 - t1, 3), "first_call_s": round(t3 - t2, 3),
                      "second_call_s": round(t4 - t3, 4), "load_to_first_result_s": round(t3 - t1, 3),
                      "checksum": float(sum(float(w.double().sum()) for w in out))}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="mixed")
    ap.add_argument("--child", nargs=2, default=None)
    args = ap.parse_args()
    if args.child:
        return child(*args.child)
    import torch, yaml
    from simwhisper_codec_amd import synth
    cfg = os.path.join(ROOT, "config", "SimWhisperCodec.yaml")
    gp = yaml.safe_load(open(cfg))["generator_params"]
    pt, pk = "/tmp/swc_synth.pt", f"/tmp/swc_synth.{args.precision}.safetensors"
    torch.save(synth.synth_state_dict(gp), pt)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_checkpoint.py"), "--config", cfg, "--in", pt, "--fold",
                        "--precision", args.precision, "--out", pk], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r.stdout.strip().splitlines()[-1])
    res = {}
    for name, ck in (("reference .pt", pt), ("packed operands", pk), ("reference .pt (2nd run, page cache warm)", pt),
                     ("packed operands (2nd run)", pk)):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg, ck], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
        print(f"{name:42s} {res[name]}")
    a, b = res["reference .pt (2nd run, page cache warm)"], res["packed operands (2nd run)"]
    if "checksum" in a and "checksum" in b:
        print("identical outputs:", a["checksum"] == b["checksum"])
    os.remove(pt); os.remove(pk)


if __name__ == "__main__":
    main()
