"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/tksmseq.h declares, fails
loudly without a GPU, and the CLI module keeps the reference's argument validation and exit codes.  No GPU."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from tksm_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "tksmseq.h")).read()
    declared = set(re.findall(r"\b(tksmseq_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.tksmseq_version()


def test_no_cpu_fallback():
    """without a usable device the product refuses to run (no routing through the oracle or any CPU path)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tksm_amd.sequence import Sequencer, TksmSeqError
    with pytest.raises(TksmSeqError) as e:
        Sequencer(0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under tksm_amd/ may import, include, link or load it"""
    bad = re.compile(r"import\s+pyoracle|from\s+oracle|from\s+pyoracle|#include\s+[\"<][^\">]*oracle|libtksm_oracle|dlopen\([^)]*oracle")
    for root, _, files in os.walk(os.path.join(ROOT, "tksm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="ignore").read()
                assert not bad.search(text), f


def _cli(*args):
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    p = subprocess.run([exe, "sequence", *args], capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


def test_cli_validation_matches_reference_messages():
    # py/sequence.py:134-164 messages; argparse usage errors exit 2, sys.exit(msg) exits 1
    assert _cli()[0] == 2
    rc, _, err = _cli("-i", "x.mdf")
    assert rc == 2 and "Must specify either --output or --perfect." in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fastq", "--badread-identity", "abc")
    assert rc == 1 and "Error: could not parse --identity values" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fastq", "--badread-identity", "101,99,5")
    assert rc == 1 and "Error: mean read identity cannot be more than 100" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fastq", "--badread-identity", "40,99,5")
    assert rc == 1 and "Error: mean read identity must be at least 50" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fastq", "--badread-identity", "95,90,5")
    assert rc == 1 and "cannot be larger than max" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fastq", "--badread-identity", "90,95,-1")
    assert rc == 1 and "Error: read identity stdev cannot be negative" in err
    rc, out, _ = _cli("--list")
    assert rc == 0 and "badread_identity" in out.split() and "input" in out.split()
    rc, _, err = _cli("-i", "x.mdf", "-O", "bam", "-o", "o.fq")
    assert rc == 2 and "invalid choice" in err


def test_cli_accepts_argparse_spellings(tmp_path):
    """The reference's parser is argparse with its defaults (py/sequence.py:35-40, :124): `--opt=value`, any unambiguous prefix of a
    long option, a value glued to a short option.  Each form reaches the same validation as the plain one; an ambiguous prefix and a
    value given to a flag are usage errors (exit 2) with argparse's wording."""
    for args in (("--input=x.mdf",), ("--inp", "x.mdf"), ("-ix.mdf",), ("--inpu=x.mdf",)):
        rc, _, err = _cli(*args)
        assert rc == 2 and "Must specify either --output or --perfect." in err, args
    for args in (("-i", "x.mdf", "--badread=o.fq", "--badread-identity=abc"), ("-ix.mdf", "-oo.fq", "--badread-id", "abc"),
                 ("-i", "x.mdf", "--perf=o.fq", "--badread-i=abc")):
        rc, _, err = _cli(*args)
        assert rc == 1 and "Error: could not parse --identity values" in err, args
    rc, _, err = _cli("-i", "x.mdf", "--bad", "o.fq")
    assert rc == 2 and "ambiguous option: --bad could match --badread, --badread-identity, --badread-error-model" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--skip-qual-compute=1")
    assert rc == 2 and "argument --skip-qual-compute: ignored explicit argument '1'" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--output-format=bam")
    assert rc == 2 and "invalid choice: 'bam'" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--nope=3")
    assert rc == 2 and "unrecognized arguments: --nope=3" in err
    rc, out, _ = _cli("--li")
    assert rc == 0 and "badread_identity" in out.split()
    # the C++ modules' parser is cxxopts (src/module.h:75-104): `--opt=value`, no abbreviations
    exe = os.path.join(ROOT, "tksm_amd", "tksm")
    p = subprocess.run([exe, "pcr", "--input=x.mdf", f"--output={tmp_path / 'o.mdf'}", "--cycles=3"], capture_output=True, text=True)
    assert p.returncode == 1 and "molecule-count is required!" in p.stderr and "Error rate is required!" in p.stderr
    p = subprocess.run([exe, "pcr", "--inp", "x.mdf"], capture_output=True, text=True)
    assert p.returncode == 1 and "Option '--inp' does not exist" in p.stderr


def test_no_output_file_is_created_before_the_run_can_start(tmp_path):
    """the outputs are created (truncated) only once devices, references, models and the input are usable: a run that stops before
    that -- here: no device, or no such input -- leaves an existing file alone and creates none"""
    import torch
    keep = tmp_path / "keep.fastq"
    keep.write_text("precious\n")
    new = tmp_path / "new.fastq"
    rc, _, err = _cli("-i", str(tmp_path / "missing.mdf"), "-o", str(keep), "--perfect", str(new))
    assert rc == 1
    assert ("no HIP device" in err) if not torch.cuda.is_available() else ("cannot open" in err)
    assert keep.read_text() == "precious\n" and not new.exists()


def test_error_model_loader_limits_are_reported(tmp_path):
    """Loader limits the reference's dict-of-lists model does not have (py/tksm_badread.py:91-117), rejected loudly with the reason
    (tksmseq_last_error(NULL) after tksmseq_prefetch_model; the same loader serves tksmseq_load_error_model): an alternative whose
    alignment puts more than 5 bases into one k-mer slot, k > 8, k-mers of unequal size.  The shipped models have <= 5 / 7 / 7."""
    import gzip
    from tksm_amd import _lib
    lib = _lib.load()

    def load(text, name):
        p = tmp_path / name
        with gzip.open(p, "wt") as f:
            f.write(text)
        rc = lib.tksmseq_prefetch_model(str(p).encode(), b"error")
        return rc, lib.tksmseq_last_error(None).decode()
    ok = "ACGTACG,0.8;ACGTTACG,0.1;ACGACG,0.05;\n"
    assert load(ok, "ok.error.gz")[0] == 0
    rc, msg = load("ACGTACG,0.8;ACGTTTTTTTACG,0.1;\n", "slot6.error.gz")
    assert rc != 0 and "alternative slot longer than 5 bases" in msg
    rc, msg = load("ACGTACGTA,0.8;ACGTACGTT,0.1;\n", "k9.error.gz")
    assert rc != 0 and "k-mer longer than 8" in msg
    rc, msg = load("ACGTACG,0.8;ACGTTACG,0.1;\nACGTAC,0.9;ACGAC,0.05;\n", "mixed.error.gz")
    assert rc != 0 and "k-mers differ in size" in msg
    rc, msg = load("ACGTACG\n", "malformed.error.gz")
    assert rc != 0 and "malformed error model line" in msg


def test_cli_utility_flags_are_validated_before_the_device_is_touched(tmp_path):
    """--batch-bytes / --in-flight / --devices / --verbosity / --log-file (src/module.h:95-122): bad values end the run
    with a message instead of spinning, being ignored or silently running on device 0"""
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--batch-bytes", "0")
    assert rc == 2 and "--batch-bytes" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--batch-bytes", "12q")
    assert rc == 2 and "--batch-bytes" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--in-flight", "0")
    assert rc == 2 and "--in-flight" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--devices", "0,x")
    assert rc == 2 and "invalid device list" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--verbosity", "LOUD")
    assert rc == 1 and "unknown verbosity level" in err
    rc, _, err = _cli("-i", "x.mdf", "-o", "o.fq", "--log-file", str(tmp_path / "no" / "such" / "dir" / "log"))
    assert rc == 1 and "cannot open log file" in err
    rc, out, _ = _cli("--list")
    assert "devices" in out.split() and "verbosity" in out.split() and "log_file" in out.split()


def test_default_model_rule_follows_the_reference(tmp_path):
    """nanopore2020 if it can be found, else `random` (py/sequence.py:86-107); all three shipped models resolve by name"""
    from tksm_amd import _lib
    lib = _lib.load()
    for name in ("nanopore2020", "nanopore2018", "pacbio2016"):
        for kind in (b"error", b"qscore"):
            assert lib.tksmseq_model_available(name.encode(), kind) == 1, (name, kind)
    assert lib.tksmseq_model_available(b"nanopore2020", b"tail") == 0          # no tail model ships with the reference
    assert lib.tksmseq_model_available(b"no_such_model", b"error") == 0




def test_read_order_is_a_stable_sort_by_length(tmp_path):
    """Every batch's read order (bucketed launches, per-range buffer sizing) is `order_by_length` of csrc/host.h, a counting sort over
    16-bit digits: the same permutation as a stable comparison sort, for empty / single / short / > 65 535-base length sets."""
    src = tmp_path / "order_check.cpp"
    src.write_text(r'''
#include "host.h"
#include <cstdio>
#include <random>
int main() {
    std::mt19937 g(5);
    const size_t ns[] = {0, 1, 2, 1000, 300000};
    const uint32_t mxs[] = {1, 7, 3000, 65536, 70000, 3000000};
    for (size_t n : ns) for (uint32_t mx : mxs) {
        std::vector<uint32_t> len(n), got, want(n);
        for (auto& v : len) v = g() % mx;
        if (n > 2) { len[0] = mx - 1; len[n - 1] = 0; }
        tkh::order_by_length(len, got);
        for (size_t i = 0; i < n; i++) want[i] = (uint32_t)i;
        std::stable_sort(want.begin(), want.end(), [&](uint32_t x, uint32_t y) { return len[x] < len[y]; });
        if (got != want) { std::printf("differs: n=%zu mx=%u\n", n, mx); return 1; }
    }
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "order_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tksm_amd", "csrc"), "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_chunk_reader_cuts_whole_molecules_and_scans_each_byte_once(tmp_path):
    """tkmod::ChunkReader (csrc/module_log.h; the same boundary search serves `tksm sequence`'s reader): pieces of about `bytes` that
    end in front of a molecule header, their concatenation is the input -- for piece sizes below a molecule's size too (a molecule
    larger than the piece size is scanned once across its refills: the search keeps the position it has covered)."""
    src = tmp_path / "chunks.cpp"
    src.write_text(r'''
#include "module_log.h"
#include <cstdio>
#include <string>
int main(int argc, char** argv) {
    std::string all, got;
    for (int m = 0; m < 400; m++) {
        all += "+mol" + std::to_string(m) + "\t1\tc=+x;\n";
        const int lines = m == 7 ? 3000 : 1 + m % 5;               // one molecule far larger than the small piece sizes
        for (int l = 0; l < lines; l++) all += "chr1\t" + std::to_string(100 * l) + "\t" + std::to_string(100 * l + 90) + "\t+\t3A,7+\n";
    }
    FILE* f = fopen(argv[1], "wb"); fwrite(all.data(), 1, all.size(), f); fclose(f);
    const uint64_t sizes[] = {64, 1000, 4096, 1 << 20};
    for (uint64_t b : sizes) {
        tkmod::ChunkReader rd; rd.in = fopen(argv[1], "rb"); rd.bytes = b;
        std::vector<char> piece; got.clear();
        size_t pieces = 0;
        while (rd.next(piece)) {
            if (piece.empty() || piece[0] != '+' || piece.back() != '\n') { std::printf("bad piece at size %llu\n", (unsigned long long)b); return 1; }
            got.append(piece.begin(), piece.end()); pieces++;
        }
        fclose(rd.in);
        if (got != all) { std::printf("differs at size %llu\n", (unsigned long long)b); return 1; }
        if (b == (1u << 20) && pieces != 1) return 2;
        if (b == 64 && pieces < 300) return 3;
    }
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "chunks"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tksm_amd", "csrc"), "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe), str(tmp_path / "in.mdf")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_prefetch_entry_points_are_host_only():
    """tksmseq_prefetch_model / tksmseq_prefetch_identity parse into the process-wide store without a context or a device"""
    from tksm_amd import _lib
    lib = _lib.load()
    models = os.path.join(ROOT, "tksm_amd", "models", "badread")
    assert lib.tksmseq_prefetch_model(os.path.join(models, "nanopore2020.error.gz").encode(), b"error") == 0
    assert lib.tksmseq_prefetch_model(os.path.join(models, "nanopore2020.error.gz").encode(), b"error") == 0     # (from the store)
    assert lib.tksmseq_prefetch_model(os.path.join(models, "nanopore2020.qscore.gz").encode(), b"qscore") == 0
    assert lib.tksmseq_prefetch_model(b"random", b"error") == 0 and lib.tksmseq_prefetch_model(b"ideal", b"qscore") == 0
    assert lib.tksmseq_prefetch_model(b"/nonexistent/model.gz", b"error") != 0
    assert lib.tksmseq_prefetch_model(b"random", b"tail") != 0
    assert lib.tksmseq_prefetch_identity(84.0, 99.0, 5.5) == 0
    assert lib.tksmseq_prefetch_identity(50.0, 60.0, 40.0) != 0          # invalid beta parameters
