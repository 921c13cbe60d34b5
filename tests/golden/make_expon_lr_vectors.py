"""Golden vectors for trainer.expon_lr from the reference's own get_expon_lr_func (utils/general_utils.py:31-60).
Run in the build container (the reference is importable there): python tests/golden/make_expon_lr_vectors.py"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from editable_gauss_refl.utils.general_utils import get_expon_lr_func  # noqa: E402

cases = [dict(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000), dict(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000),
         dict(lr_init=0.0, lr_final=0.0, max_steps=100), dict(lr_init=3e-3, lr_final=3e-3, max_steps=10)]
steps = np.array([-1, 0, 1, 7, 100, 499, 500, 501, 1999, 2000, 2001, 15000, 30000, 99999])
out = {"steps": steps, "cases": np.array([repr(c) for c in cases])}
for k, c in enumerate(cases):
    f = get_expon_lr_func(**c)
    out[f"lr{k}"] = np.array([f(int(s)) for s in steps], np.float64)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "expon_lr.npz"), **out)
print({k: v[:4] for k, v in out.items() if k.startswith("lr")})
