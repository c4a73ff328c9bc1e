"""Every ``file:line`` citation of the reference in this repository must resolve.

Round 1 shipped citations offset by +113 (src/forecasting.jl) and +20
(src/make_and_fit_model.jl); this test fails on any cited line beyond the end of the file it
names.  The line counts below are those of the reference snapshot the build was made against
(SURVEY.md section 0 lists the six source files); when the reference checkout is on the machine
(the build container; never the GPU box) the table itself is checked against it.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

LINES = {
    "Project.toml": 21,
    "README.md": 116,
    "src/NowcastAutoGP.jl": 20,
    "src/TData.jl": 74,
    "src/create_nowcast_data.jl": 76,
    "src/forecasting.jl": 167,
    "src/make_and_fit_model.jl": 93,
    "src/transformations.jl": 174,
    "test/runtests.jl": 2,
    "test/test_aqua.jl": 6,
    "test/test_forecasting.jl": 116,
    "test/test_gpconfig.jl": 43,
    "test/test_helper_functions.jl": 430,
    "test/test_model_fitting.jl": 138,
    "test/test_nowcast_functions.jl": 295,
    "docs/vignettes/getting-started.jl": 834,
    "docs/vignettes/setting-priors.jl": 403,
    "docs/vignettes/use-percentages.jl": 293,
    "docs/src/vignettes/getting-started.md": 903,
    "docs/src/vignettes/setting-priors.md": 533,
    "docs/src/vignettes/use-percentages.md": 328,
    "docs/src/index.md": 34,
}

# the driver's / judge's files quote citations of earlier rounds verbatim
SKIP = {"SURVEY.md", "VERDICT.md", "ADVICE.md", "BASELINE.md", "BASELINE.json", "PAPERS.md",
        "SNIPPETS.md", "tests/test_citations.py"}
CITE = re.compile(r"((?:src|test|docs)/[A-Za-z0-9_./-]+\.(?:jl|md)|Project\.toml|README\.md)"
                  r":(\d+(?:-\d+)?(?:, ?\d+(?:-\d+)?)*)")


def _tracked_text_files():
    out = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True)
    names = out.stdout.split() if out.returncode == 0 else []
    if not names:   # a snapshot without .git (the GPU box): walk the tree
        for d, _, fs in os.walk(ROOT):
            if any(p in d for p in ("/.git", "/gpurun_out", "/__pycache__", "/profiles")):
                continue
            names += [os.path.relpath(os.path.join(d, f), ROOT) for f in fs]
    keep = (".py", ".h", ".hip", ".md", ".c", ".jl", ".sh", ".json")
    return [n for n in names if n.endswith(keep) and n not in SKIP
            and not n.startswith(("profiles/", "gpurun_out/")) and "_r0" not in n]


def _citations():
    for name in _tracked_text_files():
        try:
            text = open(os.path.join(ROOT, name), encoding="utf-8").read()
        except (OSError, UnicodeDecodeError):
            continue
        for m in CITE.finditer(text):
            ref = m.group(1)
            # "setting-priors.md:239" style short forms are resolved against docs/src/vignettes
            for num in re.findall(r"\d+", m.group(2)):
                yield name, ref, int(num)


def test_line_table_matches_the_reference_when_present():
    if not os.path.isdir(REFERENCE):
        return
    for rel, n in LINES.items():
        with open(os.path.join(REFERENCE, rel), encoding="utf-8") as fh:
            assert sum(1 for _ in fh) == n, rel


def test_no_citation_points_beyond_the_end_of_its_file():
    bad, seen = [], 0
    for where, ref, line in _citations():
        seen += 1
        if ref not in LINES:
            bad.append(f"{where}: cites unknown reference file {ref}")
        elif not 1 <= line <= LINES[ref]:
            bad.append(f"{where}: {ref}:{line} is beyond EOF ({LINES[ref]} lines)")
    assert seen > 50, "the citation scan found almost nothing — pattern broken?"
    assert not bad, "\n".join(bad[:40])
