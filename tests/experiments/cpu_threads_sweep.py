"""Times the CPU oracle (reference torch ops) at several thread counts on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import synth, vocoder_oracle as vo

hp = vo.hifigan_v1_hp()
sd = synth.synth_state_dict(synth.hifigan_param_shapes(80, hp), 1234)
mel = synth.synth_mel(4, 80, 256, seed=2)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        vo.hifigan_forward(sd, hp, synth.synth_mel(1, 80, 32, seed=1))
        t0 = time.perf_counter(); vo.hifigan_forward(sd, hp, mel); dt = time.perf_counter() - t0
    print(f"threads={th} B=4,T=256: {dt:.2f}s  x{4*256*256/22050/dt:.2f} RT", flush=True)
