"""The three mirrors of the C ABI -- the ctypes structure of the Python front, the oracle's parameter structure and the
`#[repr(C)]` structure + `extern "C"` block of the Rust shim (shipped uncompiled: the image has no rustc) -- are checked
against include/speechsauce_amd.h by parsing the sources: field order, field types, size, and for every function the Rust shim
declares, its name, arity and argument kinds."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "speechsauce_amd.h")
SHIM = os.path.join(ROOT, "mfcc-rust_amd", "rust-shim", "src", "lib.rs")

C_TO_KIND = {"uint32_t": "u32", "int32_t": "i32", "float": "f32", "int": "i32", "long": "long", "size_t": "usize"}
RUST_TO_KIND = {"u32": "u32", "i32": "i32", "f32": "f32", "c_int": "i32", "c_long": "long", "usize": "usize"}
CTYPES_TO_KIND = {C.c_uint32: "u32", C.c_int32: "i32", C.c_float: "f32"}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def header_struct():
    text = _strip_c_comments(open(HEADER).read())
    body = re.search(r"typedef struct ss_params \{(.*?)\} ss_params;", text, flags=re.S).group(1)
    return [(C_TO_KIND[t], n) for t, n in re.findall(r"(\w+)\s+(\w+)\s*;", body)]


def header_functions():
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for ret, name, args in re.findall(r"^\s*(const char \*|int|void)\s*(ss_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S | re.M):
        kinds = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                kinds.append("ptr")
            else:
                kinds.append(C_TO_KIND[a.replace("const ", "").split()[0]])
        out[name] = (ret.strip(), kinds)
    return out


def rust_struct():
    text = open(SHIM).read()
    body = re.search(r"#\[repr\(C\)\][^{]*?pub struct SsParams \{(.*?)\n\}", text, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    return [(RUST_TO_KIND[t], n) for n, t in re.findall(r"pub (\w+):\s*(\w+)\s*,", body)]


def rust_externs():
    text = open(SHIM).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"fn (\w+)\((.*?)\)\s*(->\s*[^;]+)?;", block, flags=re.S):
        kinds = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",") if x.strip()]:
            ty = a.split(":", 1)[1].strip()
            kinds.append("ptr" if ty.startswith("*") else RUST_TO_KIND[ty])
        out[name] = (ret.replace("->", "").strip(), kinds)
    return out


def test_ctypes_structure_matches_the_header():
    import sys

    sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
    from speechsauce_amd._lib import SsParams

    want = header_struct()
    got = [(CTYPES_TO_KIND[t], n) for n, t in SsParams._fields_]
    assert got == want
    assert C.sizeof(SsParams) == 4 * len(want)  # every field is 4 bytes wide, no padding
    for i, (_, n) in enumerate(want):
        assert getattr(SsParams, n).offset == 4 * i


def test_oracle_structure_matches_the_header():
    import sys

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_c import OrcParams

    assert [(CTYPES_TO_KIND[t], n) for n, t in OrcParams._fields_] == header_struct()


def test_rust_shim_structure_matches_the_header():
    want = header_struct()
    assert rust_struct() == want
    assert 4 * len(want) == 4 * len(rust_struct())


def test_rust_shim_externs_match_the_header():
    hdr = header_functions()
    ext = rust_externs()
    assert len(ext) >= 10
    for name, (ret, kinds) in ext.items():
        assert name in hdr, f"{name} is not declared in include/speechsauce_amd.h"
        hret, hkinds = hdr[name]
        assert kinds == hkinds, (name, kinds, hkinds)
        if hret == "int":
            assert ret == "c_int", name
        elif hret == "void":
            assert ret == "", name
        else:
            assert ret.startswith("*const c_char"), name


def test_struct_size_is_announced_by_the_library(sslib):
    """ss_params_default writes struct_size = sizeof(ss_params): the compiled library agrees with the parsed header."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "mfcc-rust_amd"))
    from speechsauce_amd._lib import SsParams

    p = SsParams()
    assert sslib.ss_params_default(C.byref(p), 16000) == 0
    assert p.struct_size == C.sizeof(SsParams) == 4 * len(header_struct())


def test_mirrors_carry_the_reference_signatures_of_the_two_stage_functions():
    """processing.rs:65-76 / :179: stack_frames(signal, sample_rate, frame_length, frame_stride, filter, zero_padding) and
    power_spectrum(frames, fft_points) -- the Rust shim and the C++ mirror offer exactly these argument lists (the config forms
    are extras beside them), so a caller of the crate switches the `use` line and nothing else."""
    import re

    rs = open(os.path.join(ROOT, "mfcc-rust_amd", "rust-shim", "src", "lib.rs")).read()
    flat = re.sub(r"\s+", " ", rs)
    assert ("pub fn stack_frames(signal: ArrayView1<f32>, sample_rate: usize, frame_length: f32, frame_stride: f32, "
            "filter: Option<fn(usize) -> Array2<f32>>, zero_padding: bool) -> Array2<f32>") in flat
    assert "pub fn power_spectrum(frames: Array2<f32>, fft_points: usize) -> Array2<f32>" in flat
    ref = re.sub(r"/\*.*?\*/", "", open("/root/reference/speechsauce/src/processing.rs").read(), flags=re.S) if os.path.exists("/root/reference") else None
    if ref is not None:  # (this container only: the GPU box has no reference checkout)
        rflat = re.sub(r"\s+", " ", ref)
        assert "pub fn stack_frames( signal: ArrayView1<f32>, sample_rate: usize, frame_length: f32, frame_stride: f32, filter: Option<fn(usize) -> Array2<f32>>, zero_padding: bool, ) -> Array2<f32>" in rflat
        assert "pub fn power_spectrum(frames: Array2<f32>, fft_points: usize ) -> Array2<f32>" in rflat
    hpp = re.sub(r"\s+", " ", open(os.path.join(ROOT, "include", "speechsauce_amd.hpp")).read())
    assert ("inline Array2 stack_frames(const float *signal, std::size_t n, std::size_t sample_rate, float frame_length, float frame_stride, "
            "FrameFilter filter, bool zero_padding)") in hpp
    assert "inline Array2 power_spectrum(const Array2 &frames, std::size_t fft_points)" in hpp


def test_rust_shim_is_drop_in_by_module_path_on_paper():
    """speechsauce/src/lib.rs:2-6 declares `pub mod config / feature / functions / processing / util`; config.rs:98 derives Clone
    for SpeechConfig and :100-126 lists its public fields.  The shim (NEVER COMPILED: no rustc in the image) is checked
    textually: the module paths exist and re-export the reference's public names, SpeechConfig derives Clone over a shared handle,
    the plain-data public fields are all there, and the try_ form of stack_frames no longer asserts."""
    rs = open(SHIM).read()
    flat = re.sub(r"\s+", " ", rs)

    def mod_body(name):
        m = re.search(r"pub mod %s \{(.*?)\n\}" % name, rs, flags=re.S)
        assert m, f"pub mod {name} missing"
        return set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\b", re.sub(r"pub use super::", "", m.group(1))))

    want = {
        "config": {"SpeechConfig", "SpeechConfigBuilder"},
        "feature": {"mfcc", "mfe", "mel_spectrogram1", "mel_spectrogram2"},
        "processing": {"preemphasis", "stack_frames", "power_spectrum", "derivative_extraction", "cmvn", "cmvnw"},
        "functions": {"frequency_to_mel", "frequency_arr_to_mel", "mel_to_frequency", "mel_arr_to_frequency", "triangle", "zero_handling",
                      "stft1", "stft2"},
        "util": {"ArrayLog"},
    }
    for mod, names in want.items():
        body = mod_body(mod)
        assert names <= body, (mod, names - body)
        for n in names:  # and every re-exported name is defined at the crate root
            assert re.search(r"pub (fn|struct|trait) %s\b" % n, rs), n
    if os.path.exists("/root/reference"):  # (this container only) the lists above are the reference's public items
        ref_lib = open("/root/reference/speechsauce/src/lib.rs").read()
        assert set(re.findall(r"^pub mod (\w+);", ref_lib, flags=re.M)) == set(want)
        for mod in ("feature", "processing", "functions"):
            ref = open(f"/root/reference/speechsauce/src/{mod}.rs").read()
            assert set(re.findall(r"^pub fn (\w+)", ref, flags=re.M)) == want[mod], mod
    # Clone over a shared handle; the handle is destroyed once, by the last clone
    assert "#[derive(Clone)] pub struct SpeechConfig {" in flat
    assert "handle: Arc<Handle>," in flat and "impl Drop for Handle" in flat and "impl Drop for SpeechConfig" not in flat
    fields = re.search(r"pub struct SpeechConfig \{(.*?)\n\}", rs, flags=re.S).group(1)
    pub_fields = re.findall(r"pub (\w+):\s*([^,]+),", re.sub(r"//[^\n]*", "", fields))
    assert dict(pub_fields) == {
        "sample_rate": "usize", "window_size": "usize", "window_size_half": "usize", "frame_length": "f32", "frame_stride": "f32",
        "num_cepstral": "usize", "num_filters": "usize", "low_frequency": "f32", "high_frequency": "f32", "freq_size": "usize",
        "frame_size": "usize", "dc_elimination": "bool", "wnorm": "f32", "window": "Vec<f32>"}
    if os.path.exists("/root/reference"):
        ref = open("/root/reference/speechsauce/src/config.rs").read()
        body = re.search(r"pub struct SpeechConfig \{(.*?)\n\}", ref, flags=re.S).group(1)
        ref_fields = dict(re.findall(r"pub (\w+):\s*([^,]+),", re.sub(r"//[^\n]*", "", body)))
        plain = {k: v for k, v in ref_fields.items() if v in ("usize", "f32", "bool", "Vec<f32>")}
        assert plain == dict(pub_fields)  # every plain-data field; the plan / state objects are documented as absent
        assert set(ref_fields) - set(plain) == {"analysis_mem", "dct_handler", "fft_handler", "fft_forward", "analysis_scratch"}
    # the try_ form reports instead of asserting (round-4 advisor finding)
    body = re.search(r"pub fn try_stack_frames\(.*?\n\}", rs, flags=re.S).group(0)
    assert "assert" not in body and "SS_ERR_ARG" in body and "u32::MAX" in body


def test_rust_shim_paper_compile_guards():
    """Round 6: the shim was read line by line as rustc would.  One textual guard per defect found, plus the
    mechanical conditions of that walk.  The shim is STILL NEVER COMPILED."""
    rs = open(SHIM).read()
    code = re.sub(r"//[^\n]*", "", rs)  # comments may name the wrong forms
    flat = re.sub(r"\s+", " ", code)
    # 1. contiguous(): the view's own lifetime (ArrayView::to_slice), written out -- not a borrow of the by-value parameter
    assert "fn contiguous<'a>(signal: ArrayView1<'a, f32>) -> std::borrow::Cow<'a, [f32]>" in flat
    assert "signal.to_slice()" in flat and ".as_slice()" not in code
    # 2. ArrayLog carries the reference's two type parameters and bounds (util.rs:372-381)
    assert "pub trait ArrayLog<A: num_traits::real::Real, I: Dimension> { fn log(self) -> Array<A, I>; }" in flat
    assert "impl<A: num_traits::real::Real, I: Dimension> ArrayLog<A, I> for Array<A, I>" in flat
    cargo = open(os.path.join(os.path.dirname(os.path.dirname(SHIM)), "Cargo.toml")).read()
    assert re.search(r'^num-traits\s*=', cargo, flags=re.M) and re.search(r'^ndarray\s*=\s*"\^0\.15"', cargo, flags=re.M)
    if os.path.exists("/root/reference"):
        ref = re.sub(r"\s+", " ", open("/root/reference/speechsauce/src/util.rs").read())
        assert "pub trait ArrayLog<A: num_traits::real::Real, I: ndarray::Dimension> { fn log(self) -> Array<A, I>; }" in ref
    # 3. slices handed to ArrayView2::from_shape are plain &[f32]
    assert flat.count("ArrayView2::from_shape((1, x.len()), &x[..])") == 2 and "from_shape((1, x.len()), &x)" not in flat
    # 4. every extern declaration is called somewhere, and every call of an ss_ function has a declaration
    declared = set(rust_externs())
    body = code[code.index("\n}", code.index('extern "C" {')):]
    called = set(re.findall(r"\b(ss_\w+)\(", body))
    assert called == declared, (declared - called, called - declared)
    # 5. everything re-exported through the module paths is `pub` at the crate root
    for m in re.finditer(r"pub use super::\{(.*?)\};", code, flags=re.S):
        for name in re.findall(r"\w+", m.group(1)):
            assert re.search(r"^pub (unsafe )?(fn|struct|trait) %s\b" % name, code, flags=re.M), name
    # 6. as_standard_layout() results are let-bound (never a temporary whose pointer outlives the statement)
    assert not re.search(r"as_standard_layout\(\)\s*\.as_ptr\(\)", code)
    assert len(re.findall(r"let (x|owned) = \w+\.as_standard_layout\(\);", code)) == code.count(".as_standard_layout()")
    # 7. no private type in a public signature
    for priv in ("Handle", "Kept"):
        assert not re.search(r"pub (unsafe )?fn [^{;]*\b%s\b" % priv, flat)
    assert "NEVER COMPILED" in rs


def test_cpp_mirror_has_the_reference_fields_and_is_copyable():
    hpp = open(os.path.join(ROOT, "include", "speechsauce_amd.hpp")).read()
    for acc in ("window_size_half()", "frame_size()", "wnorm()", "window()", "freq_size()", "dc_elimination()"):
        assert acc + " const" in hpp, acc
    assert "= delete" not in hpp and "std::shared_ptr<ss_config>" in hpp
