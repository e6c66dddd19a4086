"""CPU-side checks of the C ABI: libmirhi.so loads, exports every symbol include/mirhi.h declares, and the
parts that need no GPU behave like the reference (defaults, enum order, error codes).  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "mirhi.h")).read()


def declared_functions():
    names = re.findall(r"\b(mirhi_[a-z0-9_]+)\s*\(", HEADER)
    return sorted(set(n for n in names if not n.endswith("_t")))


def test_library_exports_every_declared_symbol(mirhi):
    lib = C.CDLL(mirhi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/mirhi.h but not exported: {missing}"
    assert len(declared_functions()) >= 60
    # the ctypes binding covers the same set
    assert set(mirhi._SIGNATURES) == set(declared_functions())
    assert mirhi.lib().mirhi_abi_version() == 5 == mirhi.ABI_VERSION      # (the binding refuses a library of another ABI at load)
    assert int(re.search(r"#define MIRHI_ABI_VERSION (\d+)u", HEADER).group(1)) == 5


def test_every_entry_point_cites_the_reference():
    """Declarations name the reference interface they replace (file:line or :line within the cited file)."""
    body = HEADER[HEADER.index("/* ---- device"):]
    decls = [l for l in body.splitlines() if re.match(r"^(mirhi_result|void|uint64_t|uint32_t|int32_t|void\*|const char\*)\s+\*?\s*mirhi_", l)]
    assert len(decls) >= 55
    cites = re.findall(r"(?:\.rs)?:\d{2,4}", body)
    assert len(cites) >= 70


def test_no_gpu_means_loud_failure_not_fallback(mirhi):
    if mirhi.Device.count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mirhi.RhiError) as e:
        mirhi.Device(0)
    assert e.value.code == mirhi.ERR_NO_SUITABLE_GPU and e.value.variant == "NoSuitableGpu"
    assert "No suitable GPU found" in str(e.value)          # error.rs:21-22


def test_pipeline_defaults_match_reference(mirhi):
    """GraphicsPipelineBuilder::new (pipeline.rs:645-698; tests pipeline.rs:1177-1189,1230-1254)."""
    d = mirhi.GraphicsPipelineBuilder().desc
    assert d.topology == mirhi.PrimitiveTopology.TriangleList
    assert d.polygon_mode == mirhi.PolygonMode.Fill
    assert d.cull_mode == mirhi.CullMode.Back
    assert d.front_face == mirhi.FrontFace.CounterClockwise
    assert d.depth_test_enable == 1 and d.depth_write_enable == 1 and d.depth_compare_op == mirhi.CompareOp.Less
    assert d.rasterization_samples == 1 and d.blend_enable == 0 and d.depth_clamp_enable == 0
    assert d.vertex_program == mirhi.Program.NONE and d.fragment_program == mirhi.Program.NONE
    assert d.color_attachment_count == 0 and d.depth_attachment_format == mirhi.Format.UNDEFINED


def test_rendering_info_defaults_match_reference(mirhi):
    """ColorAttachment::new / DepthAttachment::new (rendering.rs:102-115,356-370; tests :1027-1073)."""
    info = mirhi.RenderingInfo()
    mirhi.lib().mirhi_rendering_info_default(C.byref(info))
    assert info.color_load_op == mirhi.LoadOp.CLEAR and info.color_store_op == mirhi.StoreOp.STORE
    assert list(info.clear_color) == [0.0, 0.0, 0.0, 1.0]
    assert info.depth_load_op == mirhi.LoadOp.CLEAR and info.depth_store_op == mirhi.StoreOp.DONT_CARE
    assert info.clear_depth == 1.0


def test_enum_order_matches_reference(mirhi):
    assert [mirhi.BufferUsage.Vertex, mirhi.BufferUsage.Index, mirhi.BufferUsage.Uniform, mirhi.BufferUsage.Storage,
            mirhi.BufferUsage.Staging, mirhi.BufferUsage.Indirect] == list(range(6))        # buffer.rs:47-60
    assert [mirhi.CompareOp.Never, mirhi.CompareOp.Less, mirhi.CompareOp.Equal, mirhi.CompareOp.LessOrEqual,
            mirhi.CompareOp.Greater, mirhi.CompareOp.NotEqual, mirhi.CompareOp.GreaterOrEqual,
            mirhi.CompareOp.Always] == list(range(8))                                        # pipeline.rs:375-386
    assert [mirhi.CullMode.NONE, mirhi.CullMode.Front, mirhi.CullMode.Back, mirhi.CullMode.FrontAndBack] == list(range(4))
    names = [mirhi.lib().mirhi_result_name(i).decode() for i in range(11)]
    assert names == ["Ok", "VulkanError", "LoadingError", "AllocatorError", "NoSuitableGpu", "ShaderError", "SurfaceError",
                     "SwapchainError", "InvalidHandle", "PipelineError", "LockPoisoned"]        # error.rs:6-50
    assert mirhi.MAX_FRAMES_IN_FLIGHT == 2                                                     # renderer/src/lib.rs:43


def test_null_handles_are_errors_not_crashes(mirhi):
    L = mirhi.lib()
    assert L.mirhi_device_wait_idle(None) == mirhi.ERR_INVALID_HANDLE
    assert L.mirhi_buffer_destroy(None) == mirhi.ERR_INVALID_HANDLE
    assert L.mirhi_cmd_begin(None) == mirhi.ERR_INVALID_HANDLE
    assert L.mirhi_fence_wait(None, 0) == mirhi.ERR_INVALID_HANDLE
    assert b"null" in L.mirhi_last_error_message()
    assert L.mirhi_buffer_size(None) == 0


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    pkg = os.path.join(ROOT, "renderer-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_binding" not in text and "libmirhi_oracle" not in text and "mirhi_oracle.h" not in text, f


def test_generated_rust_sys_crate_covers_the_header():
    """bindings/rust/mirhi-sys/src/lib.rs (tools/gen_rust_sys.py; no Rust toolchain here, so it is generated, not compiled):
    up to date with the header, one `pub fn` per C function with the same number of parameters, every struct and enum."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen", os.path.join(root, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, funcs = gen.generate()
    assert open(gen.OUT).read() == text, "stale: run tools/gen_rust_sys.py"
    opaque, enums, structs, parsed = gen.parse(open(gen.HEADER).read())
    assert len(parsed) >= 60 and {"mirhi_pipeline_desc", "mirhi_rendering_info", "mirhi_viewport", "mirhi_rect2d"} <= set(structs)
    for name, ret, params in parsed:
        m = re.search(r"pub fn " + name + r"\((.*?)\)( -> [^;]+)?;", text)
        assert m, name
        assert (len([p for p in m.group(1).split(",") if p.strip()]) == len(params)), name
    for e, items in enums.items():
        for k, v in items:
            assert f"pub const {k}: {e} = {v};" in text
    assert "pub const MIRHI_ERR_PIPELINE: mirhi_result = 9;" in text


def test_rust_wrapper_crate_only_uses_what_the_sys_crate_declares():
    """bindings/rust/renderer-rhi-hip (the safe wrappers with the rhi names; not compiled here -- no Rust toolchain) must stay
    in step with the generated mirhi-sys crate: every function, constant, struct and struct field it names exists there."""
    import glob
    sys_src = open(os.path.join(ROOT, "bindings", "rust", "mirhi-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (mirhi_\w+)", sys_src)) | set(re.findall(r"pub const (MIRHI_\w+)", sys_src)) \
        | set(re.findall(r"pub struct (mirhi_\w+)", sys_src)) | set(re.findall(r"pub type (mirhi_\w+)", sys_src))
    fields = {}
    for name, body in re.findall(r"pub struct (mirhi_\w+) \{\n(.*?)\n\}", sys_src, re.S):
        fields[name] = set(re.findall(r"pub (\w+):", body))
    files = sorted(glob.glob(os.path.join(ROOT, "bindings", "rust", "renderer-rhi-hip", "src", "*.rs")))
    assert len(files) >= 8
    used = set()
    for f in files:
        src = open(f).read()
        used |= set(re.findall(r"mirhi_sys::(mirhi_\w+|MIRHI_\w+)", src))
        if "use mirhi_sys::*;" in src:
            used |= set(re.findall(r"\b(MIRHI_[A-Z_]+)\b", src))
        assert src.count("{") == src.count("}") and src.count("(") == src.count(")"), f"unbalanced delimiters in {f}"
    missing = sorted(u for u in used if u not in declared)
    assert not missing, f"renderer-rhi-hip names symbols mirhi-sys does not declare: {missing}"
    assert len(used) > 60
    pipe = open(os.path.join(ROOT, "bindings", "rust", "renderer-rhi-hip", "src", "pipeline.rs")).read()
    for fld in set(re.findall(r"self\.desc\.(\w+)", pipe)):
        assert fld in fields["mirhi_pipeline_desc"], fld
    cmd = open(os.path.join(ROOT, "bindings", "rust", "renderer-rhi-hip", "src", "command.rs")).read()
    for fld in set(re.findall(r"\bri\.(\w+) =", cmd)):
        assert fld in fields["mirhi_rendering_info"], fld


def test_frame_loop_library_exports_every_declared_symbol(mirhi, tmp_path):
    """include/mirhost.h (the reference's draw-submit loop as native host code over the C ABI): libmirhost.so loads next to libmirhi.so and exports
    every function the header declares; every declaration names what it mirrors in crates/renderer."""
    from renderer_rs_amd import frameloop
    text = open(os.path.join(ROOT, "include", "mirhost.h")).read()
    body = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    names = re.findall(r"\b(mirhost_\w+)\s*\(", body)
    assert set(names) >= {"mirhost_frame_loop_create", "mirhost_frame_loop_run", "mirhost_frame_loop_last_image", "mirhost_frame_loop_phase_seconds",
                          "mirhost_frame_loop_destroy", "mirhost_last_error_message"}
    lib = C.CDLL(frameloop.build())
    missing = [n for n in set(names) if not hasattr(lib, n)]
    assert not missing, f"declared in include/mirhost.h but not exported: {missing}"
    assert "renderer.rs:367-449" in text and "frame_manager.rs:299-539" in text
    # the ctypes mirror of the two structs has the C layout gcc gives the header
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mirhost.h"\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(mirhost_draw), '
                   'sizeof(mirhost_frame_desc), offsetof(mirhost_draw, count), offsetof(mirhost_frame_desc, draws)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(frameloop.HostDraw), C.sizeof(frameloop.FrameDesc), frameloop.HostDraw.count.offset, frameloop.FrameDesc.draws.offset]
