#!/usr/bin/env python
"""Golden vectors for UpSample1d / DownSample1d / LowPassFilter1d / Snake / SnakeBeta / a ratio-3 Activation1d from the
REAL reference classes (modules/anti_aliasing/{resample,filter,act}.py, modules/activation_functions/snake.py), run on CPU in the build container (needs /root/reference):
    python tests/golden/make_golden_resample.py   ->  tests/golden/golden_resample.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs)


def main():
    mg.install_stubs()
    sys.path.insert(0, mg.REF)
    from modules.anti_aliasing.filter import LowPassFilter1d
    from modules.anti_aliasing.resample import DownSample1d, UpSample1d

    g = torch.Generator().manual_seed(17)
    out = {}
    x = torch.randn(2, 3, 37, generator=g)
    out["x"] = x.numpy()
    with torch.no_grad():
        for ratio, ks in [(2, None), (3, None), (4, 16), (2, 7)]:
            tag = f"r{ratio}k{ks or 0}"
            up, dn = UpSample1d(ratio, ks), DownSample1d(ratio, ks)
            out[f"up_{tag}_filter"] = up.filter.reshape(-1).numpy()
            out[f"up_{tag}_y"] = up(x).numpy()
            out[f"down_{tag}_filter"] = dn.lowpass.filter.reshape(-1).numpy()
            out[f"down_{tag}_y"] = dn(x).numpy()
        for tag, kw in {"lp_k12": dict(cutoff=0.25, half_width=0.3, kernel_size=12),
                        "lp_k9_s2_nopad": dict(cutoff=0.2, half_width=0.3, stride=2, padding=False, kernel_size=9),
                        "lp_k8_reflect": dict(cutoff=0.3, half_width=0.4, padding_mode="reflect", kernel_size=8),
                        "lp_k5_zeros": dict(cutoff=0.4, half_width=0.5, padding_mode="constant", kernel_size=5)}.items():
            lp = LowPassFilter1d(**kw)
            out[f"{tag}_filter"] = lp.filter.reshape(-1).numpy()
            out[f"{tag}_y"] = lp(x).numpy()
        x1 = torch.randn(1, 2, 1, generator=g)           # a single sample: everything is padding
        out["x1"] = x1.numpy()
        out["up_T1_y"] = UpSample1d(2)(x1).numpy()
        out["down_T1_y"] = DownSample1d(2)(x1).numpy()
        from modules.activation_functions.snake import Snake, SnakeBeta
        from modules.anti_aliasing.act import Activation1d

        for tag, cls, log in [("snake_lin", Snake, False), ("snake_log", Snake, True), ("snakebeta_lin", SnakeBeta, False),
                              ("snakebeta_log", SnakeBeta, True)]:
            act = cls(3, alpha_logscale=log)
            act.alpha.data = torch.randn(3, generator=g) * 0.4 + (0.0 if log else 1.0)
            out[f"{tag}_alpha"] = act.alpha.data.numpy().copy()
            if cls is SnakeBeta:
                act.beta.data = torch.randn(3, generator=g) * 0.4 + (0.0 if log else 1.0)
                out[f"{tag}_beta"] = act.beta.data.numpy().copy()
            out[f"{tag}_y"] = act(x).numpy()
        a3 = Activation1d(SnakeBeta(3, alpha_logscale=True), up_ratio=3, down_ratio=3, up_kernel_size=18, down_kernel_size=18)
        a3.act.alpha.data = torch.randn(3, generator=g) * 0.3
        a3.act.beta.data = torch.randn(3, generator=g) * 0.3
        out["act_r3_alpha"] = a3.act.alpha.data.numpy().copy()
        out["act_r3_beta"] = a3.act.beta.data.numpy().copy()
        out["act_r3_y"] = a3(x).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_resample.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
