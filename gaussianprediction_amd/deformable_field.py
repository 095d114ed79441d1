"""Deformable_Field: same constructor, parameter names and state_dict keys as the reference module
[REF scene/deformable_field.py:74-127] (`mlp.{0,2,4,6}.{weight,bias}`, `feature_to_deformation.0.*`), so
reference checkpoints (`gaussians.state_dict()`, [REF train.py:199-201]) load unchanged and
`self.df_model.parameters()` feeds the same optimizer group "df_mlp" [REF scene/gaussian_model.py:406].
The forward is the fused HIP kernel over those same Parameters (fp32 matrix cores) for the reference's operating point
(d=4, w=256, split_xyz=False, use_softmax=False: options/gaussian_option.py:54-55, scene/gaussian_model.py:79).  Every other
depth / width and the two dormant variants (use_softmax, split_xyz -- same ModuleDict names, so their state_dicts load too) run
layer by layer on the library's generic dense-layer entries (deform_ops.GenericMlp: exact fp32, any sizes; csrc/deform_generic.hip).
"""
from __future__ import annotations

import torch
from torch import nn

from .deform_ops import FusedMlp, FusedMlp16, GenericMlp, MlpInput


class Deformable_Field(nn.Module):
    def __init__(self, input_dim, output_dim=10, d=8, w=256, use_softmax=False, split_xyz=False, precision="fp32", range_guard="lazy"):
        """`precision` (extension): "fp32" = exact-fp32 matrix cores (default); "fp32s" = fp32 operands carried as fp16 (hi, lo')
        pairs on the 16-bit matrix cores (22 significant bits, fp32 accumulation: meets the fp32 kernels' test bars at ~2x their speed;
        |activations| < 65504); "fp16" / "bf16" = 16-bit operands with fp32 accumulation, ~16x the MFMA rate (BASELINE config 5).
        The 16-bit kernels serve calls with more than 2048 rows; smaller ones always use the exact-fp32 small-row kernels.
        `range_guard` ("fp32s" only): the (hi, lo') form saturates at the fp16 range, silently; the split kernel therefore raises a
        device flag when a hidden activation reaches 2^15.  "sync": the flag is read after every pass (one host synchronisation)
        and a flagged pass is REPEATED on the exact-fp32 kernels; "lazy" (default): the flag travels to the host asynchronously and
        is looked at when the next pass starts -- a flagged model switches to the exact-fp32 kernels from then on, with a warning
        (the flagged pass itself ran saturated: finite, and wrong where an activation exceeded 65504); "off": no flag."""
        super().__init__()
        if precision not in ("fp32", "fp32s", "fp16", "bf16"):
            raise ValueError("precision must be fp32, fp32s, fp16 or bf16")
        self.precision = precision
        if range_guard not in ("lazy", "sync", "off"):
            raise ValueError("range_guard must be lazy, sync or off")
        self.range_guard = range_guard
        self.range_tripped = False            # an activation left the split form's range: the exact-fp32 kernels from now on
        self._flag = self._flag_host = self._flag_event = None
        if d < 1 or w < 1:
            raise ValueError("Deformable_Field needs d >= 1 hidden layers of width w >= 1")
        self.input_dim, self.output_dim, self.d, self.w = input_dim, output_dim, d, w
        self.use_softmax, self.split_xyz = use_softmax, split_xyz
        # the fused kernels implement d = 4, w = 256 without the dormant variants; anything else takes the generic layers
        self.generic = bool(split_xyz or use_softmax or d != 4 or w != 256)

        def hidden():
            layers = []
            for i in range(d):
                layers.append(nn.Linear(input_dim if i == 0 else w, w))
                layers.append(nn.ReLU())
            return nn.Sequential(*layers)
        if split_xyz:                    # one network per output channel, each with ONE output [REF scene/deformable_field.py:84-99]
            self.output_times, self.output_dim = output_dim, 1
            self.mlp = nn.ModuleDict({f"mlp{k:d}": hidden() for k in range(self.output_times)})
            self.feature_to_deformation = nn.ModuleDict({f"feature_to_deformation{k:d}": nn.Sequential(nn.Linear(w, 1))
                                                         for k in range(self.output_times)})
        else:
            self.mlp = hidden()
            self.feature_to_deformation = nn.Sequential(nn.Linear(w, output_dim))

    def _chain(self, mlp, head):
        wb = []
        for i in range(self.d):
            wb += [mlp[2 * i].weight, mlp[2 * i].bias]
        return wb + [head[0].weight, head[0].bias]

    def _generic(self, x):
        """[REF scene/deformable_field.py:112-127]"""
        if self.split_xyz:
            outs = [GenericMlp.apply(x, self.use_softmax, *self._chain(self.mlp[f"mlp{k:d}"], self.feature_to_deformation[f"feature_to_deformation{k:d}"]))
                    for k in range(self.output_times)]
            return torch.cat(outs, dim=-1)
        return GenericMlp.apply(x, self.use_softmax, *self._chain(self.mlp, self.feature_to_deformation))

    def _wb(self):
        return self._chain(self.mlp, self.feature_to_deformation)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [M, input_dim] already-concatenated input (the reference's call form)."""
        if self.generic:
            return self._generic(x)
        return FusedMlp.apply(x, None, None, 0, 0, *self._wb())

    def forward_fused(self, feature, xyz, t, xyz_freq, time_freq) -> torch.Tensor:
        """Fused form used by GaussianModel: builds [feature | PE(xyz) | PE(t)] inside the kernel
        (get_motion_delta, REF scene/gaussian_model.py:180-184) -- the [M, input_dim] input and the
        [M,256] activations never touch HBM in inference."""
        if self.generic:
            return self._generic(MlpInput.apply(feature, xyz, t, xyz_freq, time_freq))
        # 16-bit operands pay off for the per-Gaussian passes (10^5..10^6 rows).  A few hundred rows (the keypoints of
        # stage 2/3) are a latency problem, for which the 16-row fp32 kernels are both faster and exact.
        if self.precision != "fp32" and feature.shape[0] > 2048 and not self.range_tripped:
            guard = self.range_guard if self.precision == "fp32s" else "off"
            flag = self._range_flag(feature.device) if guard != "off" else None
            if flag is not None and self.range_tripped:                       # (the lazy check just tripped)
                return FusedMlp.apply(feature, xyz, t, xyz_freq, time_freq, *self._wb())
            out = FusedMlp16.apply(feature, xyz, t, xyz_freq, time_freq, self.precision, flag, *self._wb())
            if guard == "sync":
                if int(flag.item()) != 0:                                     # this pass saturated: repeat it exactly
                    self._trip("this pass was repeated on the exact-fp32 kernels")
                    return FusedMlp.apply(feature, xyz, t, xyz_freq, time_freq, *self._wb())
            elif guard == "lazy":
                self._flag_host.copy_(flag, non_blocking=True)
                self._flag_event = torch.cuda.Event()
                self._flag_event.record()
            return out
        return FusedMlp.apply(feature, xyz, t, xyz_freq, time_freq, *self._wb())

    def _trip(self, what):
        import warnings
        self.range_tripped = True
        warnings.warn("Deformable_Field(precision='fp32s'): a hidden activation reached 2^15 (the split-fp16 form saturates at 65504); "
                      + what + "; every later pass uses them too", RuntimeWarning, stacklevel=3)

    def _range_flag(self, device):
        """The device word of the range guard (created on first use); in lazy mode first looks at what the PREVIOUS pass left."""
        if self._flag is None or self._flag.device != device:
            self._flag = torch.zeros(1, dtype=torch.int32, device=device)
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._flag_event = None
        if self.range_guard == "lazy" and self._flag_event is not None and self._flag_event.query():
            self._flag_event = None
            if int(self._flag_host[0]) != 0:
                self._trip("the pass that raised the flag ran saturated; from this pass on the exact-fp32 kernels run")
        return self._flag
