"""Sanitizer builds of the host layer (race / memory coverage of the code
behind the C-ABI).  ngp_api.hip and the launchers of ngp_kernels.hip are compiled host-only with
ThreadSanitizer, and again with AddressSanitizer + UBSan, linked against a mock HIP runtime
(tests/sanitize/mock_hip.cpp: zeroed host memory for device memory, inert streams and events,
launches that do nothing) and driven by tests/sanitize/host_stress.cpp: four threads on one context
plus one thread creating and destroying contexts, specs flipped between calls, staged jobs, resident
factors, error returns — and by tests/sanitize/combine_stress.cpp: bursts of concurrent one-shot
callers behind a "busy device" (the mock's synchronisation sleeps), i.e. the flat combining of
include/ngp.h "concurrent callers": no lost wake-up, every caller's arrays written, a burst costs
about one call's launches.  No GPU is involved (GPU sanitizers are not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nowcastautogp_amd", "csrc")
SAN = os.path.join(ROOT, "tests", "sanitize")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


def build(tmp, flags, tag, driver="host_stress"):
    objs = []
    for src in ("ngp_api", "ngp_kernels"):
        o = os.path.join(tmp, f"{src}_{tag}.o")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g",
                               "-std=c++17", "-w", *flags, "-c", os.path.join(CSRC, src + ".hip"),
                               "-o", o])
        objs.append(o)
    # the registration code of a host-only object still names its (absent) device image
    nm = subprocess.check_output(["nm", "-u", *objs], text=True)
    fat = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin" in ln})
    stub = os.path.join(tmp, f"fatbin_{tag}.c")
    with open(stub, "w") as f:
        for name in fat:
            f.write(f"const char {name}[16] = {{0}};\n")
    clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    for src in (os.path.join(SAN, "mock_hip.cpp"), os.path.join(SAN, driver + ".cpp"), stub):
        o = os.path.join(tmp, os.path.basename(src).rsplit(".", 1)[0] + f"_{tag}.o")
        lang = ["-x", "c"] if src.endswith(".c") else ["-std=c++17"]
        subprocess.check_call([clangxx, *lang, "-O1", "-g", "-w", *flags, "-c", src, "-o", o])
        objs.append(o)
    exe = os.path.join(tmp, f"{driver}_{tag}")
    subprocess.check_call([clangxx, *flags, *objs, "-lpthread", "-o", exe])
    return exe


@pytest.mark.skipif(HIPCC is None, reason="no hipcc")
@pytest.mark.parametrize("tag,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0"}),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
     {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"}),
])
@pytest.mark.parametrize("driver", ["host_stress", "combine_stress"])
def test_host_layer_under_sanitizers(tmp_path, tag, flags, env, driver):
    exe = build(str(tmp_path), flags, tag, driver)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env={**os.environ, **env})
    report = out.stdout[-3000:] + out.stderr[-6000:]
    assert "ThreadSanitizer" not in out.stderr, report
    assert "AddressSanitizer" not in out.stderr and "LeakSanitizer" not in out.stderr, report
    assert "runtime error" not in out.stderr, report
    assert out.returncode == 0, report
    assert "0 failures" in out.stdout, report
