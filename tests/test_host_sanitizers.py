"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer: tests/support/host_asan.c compiles the host sources
into one instrumented driver (no GPU call is made) and feeds generated input to what runs on untrusted text: the
highlighter (tokenizer, matcher, mark assembly, snippets) and the dictionary export behind the device term lookup."""
import os
import subprocess

from util import ROOT


def test_highlight_and_dictionary_export_are_clean(tmp_path):
    exe = str(tmp_path / "host_asan")
    host = os.path.join(ROOT, "marlin_amd", "host")
    srcs = [os.path.join(ROOT, "tests", "support", "host_asan.c")] + \
           [os.path.join(host, f) for f in ("mhighlight.c", "mdict.c", "mquery.c", "mshard.c", "manalyzer.c")]
    subprocess.run(["gcc", "-std=gnu11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-static-libasan", "-static-libubsan", "-I" + os.path.join(ROOT, "include"), "-I" + host] + srcs + ["-o", exe, "-lm", "-lpthread"], check=True)
    # (the sanitizer runtimes are linked statically: the environment, preloads included, is passed on as it is)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout[-3000:]
