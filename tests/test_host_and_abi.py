"""CPU: host-side logic and the C-ABI surface (library loads, exports every declared symbol,
argument validation happens before any kernel launch, product fails loudly without a GPU)."""
import ctypes as C
import os
import re

import pytest
import torch

import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gp_hip.h")).read()
    declared = set(re.findall(r"\b(gp_[a-z_0-9]+)\s*\(", hdr)) - {"gp_alloc_fn"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    l = _lib.lib()
    for name in declared:
        assert hasattr(l, name), name
    assert b"gfx950" in l.gp_version()


def test_struct_sizes_match_header_layout():
    assert C.sizeof(_lib.RasterSettingsC) == 9 * 4 + 4 + 4 * 8 + 8 + 8 + 8 + 4 + 4 + 8 + 4 + 4   # 9 x 4-byte + pad + 4 pointers + capacity + status pointer + event + depth-key bits, base, range pointer + raw_activations + reserved
    assert C.sizeof(_lib.RasterInputsC) == 9 * 8
    assert C.sizeof(_lib.RasterSavedC) == 7 * 8
    assert C.sizeof(_lib.RasterGradsC) == 9 * 8 + 8 + 8     # + accumulate_shs (padded) + adam_shs
    assert C.sizeof(_lib.AdamFuseC) == 4 * 8 + 5 * 4 + 4 + 8 + 8   # 4 pointers, 5 floats + pad, step, skip_flag
    assert C.sizeof(_lib.MlpParamsC) == 4 * 4 + 12 * 8      # 4 dims, w[5], b[5], packed, scratch (round 6)
    assert C.sizeof(_lib.MlpInputC) == 8 + 3 * 4 + 4 + 3 * 8
    assert C.sizeof(_lib.BlendArgsC) == 2 * 8 + 3 * 4 + 4 + 6 * 8


def test_mlp_scratch_size_query():
    """gp_mlp_scratch_bytes (round 6): 8 KB of counters (32 row tiles x 128 B + the error word at byte 4096) + the exchange region of
    four hidden layers; 0 where the feature-split kernels do not run (more than 512 rows)."""
    l = _lib.lib()
    assert int(l.gp_mlp_scratch_bytes(C.c_int64(250))) == 8192 + 4 * 250 * 256 * 4
    assert int(l.gp_mlp_scratch_bytes(C.c_int64(512))) == 8192 + 4 * 512 * 256 * 4
    assert int(l.gp_mlp_scratch_bytes(C.c_int64(513))) == 0 and int(l.gp_mlp_scratch_bytes(C.c_int64(0))) == 0


def test_abi_validation_errors_are_reported_not_thrown():
    l = _lib.lib()
    st = _lib.RasterSettingsC(0, 0, 1.0, 1.0, 1.0, 3, 16, 0, 0, None, None, None, None)
    inp = _lib.RasterInputsC(0, None, None, None, None, None, None, None, None)
    out = _lib.RasterOutputsC(None, None, None, None)
    saved = _lib.RasterSavedC()
    alloc = _lib.TorchAllocator("cpu")
    rc = l.gp_raster_forward(C.byref(st), C.byref(inp), C.byref(out), C.byref(saved), alloc.cb, None, None)
    assert rc != 0 and b"image size" in l.gp_last_error()
    st.image_height, st.image_width = 16, 16
    inp.num_gaussians = 4
    rc = l.gp_raster_forward(C.byref(st), C.byref(inp), C.byref(out), C.byref(saved), alloc.cb, None, None)
    assert rc != 0 and b"exactly one of either SHs" in l.gp_last_error()
    p = _lib.MlpParamsC(104, 128, 4, 7)
    x = _lib.MlpInputC(0, 32, 10, 6, None, None, None)
    rc = l.gp_mlp_forward(C.byref(p), C.byref(x), None, None, None)
    assert rc != 0 and b"d=4, w=256" in l.gp_last_error()
    with pytest.raises(_lib.GpHipError):
        _lib.check(rc, "gp_mlp_forward")


def test_rasterizer_argument_contract_and_no_cpu_fallback():
    bg = torch.zeros(3)
    rs = gpa.GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=bg,
                                           scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                           sh_degree=3, campos=torch.zeros(3), prefiltered=False)
    assert rs.debug is False                      # reference does not pass debug (gaussian_renderer/__init__.py:49)
    r = gpa.GaussianRasterizer(raster_settings=rs)
    m3 = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m3, means2D=m3, opacities=torch.ones(4, 1), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m3, means2D=m3, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3))
    # CPU tensors: the product must fail loudly, never silently fall back
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=m3, means2D=m3, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))


def test_deformable_field_state_dict_keys_match_reference():
    net = gpa.Deformable_Field(104, output_dim=7, d=4, w=256, split_xyz=False)
    keys = list(net.state_dict().keys())
    assert keys == ["mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias", "mlp.4.weight", "mlp.4.bias",
                    "mlp.6.weight", "mlp.6.bias", "feature_to_deformation.0.weight", "feature_to_deformation.0.bias"]
    assert sum(p.numel() for p in net.parameters()) == 226055
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(3, 104))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gaussianprediction_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gp_oracle" not in src.replace("oracle/gp_oracle.c", ""), f


def test_capacity_mode_settings_are_validated_on_the_host():
    """GaussianRasterizationSettings.binning_capacity / binning_status (extension): defaults keep the reference's
    11-kwarg construction working; a capacity without a status word, or a malformed status tensor, is rejected before
    anything is launched."""
    import torch
    from gaussianprediction_amd import rasterizer as R
    dev = torch.device("cpu")
    kw = dict(image_height=8, image_width=8, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3), scale_modifier=1.0,
              viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=3, campos=torch.zeros(3), prefiltered=False)
    rs = R.GaussianRasterizationSettings(**kw)
    assert rs.debug is False and rs.binning_capacity == 0 and rs.binning_status is None
    st, keep = R._settings_c(rs, dev, 16)
    assert st.binning_capacity == 0 and not st.binning_status
    status = torch.zeros(2, dtype=torch.int32)
    st, keep = R._settings_c(R.GaussianRasterizationSettings(**kw, binning_capacity=1000, binning_status=status), dev, 16)
    assert st.binning_capacity == 1000 and st.binning_status == status.data_ptr() and any(k is status for k in keep)
    with pytest.raises(RuntimeError, match="needs binning_status"):
        R._settings_c(R.GaussianRasterizationSettings(**kw, binning_capacity=1000), dev, 16)
    with pytest.raises(RuntimeError, match="int32"):
        R._settings_c(R.GaussianRasterizationSettings(**kw, binning_capacity=1000, binning_status=torch.zeros(2)), dev, 16)


# ---- one layout, three statements of it: include/gp_hip.h (gcc), gaussianprediction_amd/_lib.py, INTEGRATION.md section 3 ----
_STRUCTS = {"gp_raster_settings": "RasterSettingsC", "gp_raster_inputs": "RasterInputsC", "gp_raster_outputs": "RasterOutputsC",
            "gp_raster_saved": "RasterSavedC", "gp_raster_grads": "RasterGradsC", "gp_adam_fuse": "AdamFuseC",
            "gp_mlp_params": "MlpParamsC", "gp_mlp16_params": "Mlp16ParamsC", "gp_mlp_grads": "MlpGradsC",
            "gp_mlp_input": "MlpInputC", "gp_blend_args": "BlendArgsC", "gp_profile_entry": "ProfileEntryC",
            "gp_step_plan": "StepPlanC", "gp_step_view": "StepViewC", "gp_step_update": "StepUpdateC"}


def _header_layout(tmp_path):
    """{struct: (sizeof, [(field, offset, size)])} of the header, measured by a C program gcc builds from it."""
    hdr = open(os.path.join(ROOT, "include", "gp_hip.h")).read()
    body = []
    for cname in _STRUCTS:
        m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S)
        assert m, cname
        text = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        fields = []
        for decl in text.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.sub(r"\[.*\]", "", part.strip().split()[-1]).lstrip("*")
                fields.append(name)
        body.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f in fields:
            body.append('printf(" %s:%%zu:%%zu", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (f, cname, f, cname, f))
        body.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gp_hip.h"\nint main(void){%s return 0;}\n' % "".join(body))
    exe = tmp_path / "layout"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = {}
    for line in subprocess.check_output([str(exe)]).decode().splitlines():
        toks = line.split()
        out[toks[0]] = (int(toks[1]), [(t.split(":")[0], int(t.split(":")[1]), int(t.split(":")[2])) for t in toks[2:]])
    return out


def _ctypes_layout(cls):
    return C.sizeof(cls), [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, *_ in cls._fields_]


def test_binding_structs_match_the_header_field_by_field(tmp_path):
    hdr = _header_layout(tmp_path)
    for cname, pyname in _STRUCTS.items():
        cls = getattr(_lib, pyname, None)
        if cls is None:
            continue
        assert _ctypes_layout(cls) == hdr[cname], (cname, _ctypes_layout(cls), hdr[cname])


def test_integration_stub_matches_header_and_binding(tmp_path):
    """INTEGRATION.md section 3 is what an integrator copies: its struct definitions must be the header's (round-2 verdict:
    the stub had fallen two fields behind, so the library would have read past the caller's struct)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 3."):md.index("## 4.")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    defs = code[:code.index("# ---- call")]
    assert "gp_raster_forward" in code and "gp_raster_backward" in code
    ns = {}
    exec(defs.replace('lib = C.CDLL("gaussianprediction_amd/libgp_hip.so")', "lib = None"), ns)
    hdr = _header_layout(tmp_path)
    checked = 0
    for cname, pyname in _STRUCTS.items():
        if pyname in ns:
            assert _ctypes_layout(ns[pyname]) == hdr[cname], (cname, _ctypes_layout(ns[pyname]), hdr[cname])
            assert _ctypes_layout(ns[pyname]) == _ctypes_layout(getattr(_lib, pyname))
            checked += 1
    assert checked == 5
    # the call part constructs every struct by keyword, with field names that exist
    import ast
    seen = set()
    for node in ast.walk(ast.parse(code[code.index("# ---- call"):])):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in ns and hasattr(ns[node.func.id], "_fields_"):
            assert not node.args, f"{node.func.id} constructed positionally"
            assert {k.arg for k in node.keywords} <= {n for n, *_ in ns[node.func.id]._fields_}, node.func.id
            seen.add(node.func.id)
    assert seen == {"RasterSettingsC", "RasterInputsC", "RasterOutputsC", "RasterSavedC", "RasterGradsC"}
