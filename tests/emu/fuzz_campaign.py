"""TEST INFRASTRUCTURE ONLY: a longer, randomly seeded run of the property behind tests/test_emu_property.py (random ragged
exact-GP problems through the whole C-ABI path on the CPU build of the kernel sources, against the oracle) - the CPU
suite runs 150 derandomised examples, this runs as many as asked for with fresh seeds and prints the first
counter-example.       python tests/emu/fuzz_campaign.py [examples] [n_max] [seed]
With FUZZ_ASAN=1 (and the sanitizer runtime preloaded: LD_PRELOAD=$(python -c "import build_emu; print(build_emu.sanitizer_runtime('asan'))")
ASAN_OPTIONS=detect_leaks=0) the campaign runs through the AddressSanitizer build: every access of the random ragged
shapes is bounds-checked."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from hypothesis import HealthCheck, given, seed, settings  # noqa: E402
from inject import installed  # noqa: E402
from problem_gen import check_problem, problems  # noqa: E402

examples = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_max = int(sys.argv[2]) if len(sys.argv) > 2 else 260
rseed = int(sys.argv[3]) if len(sys.argv) > 3 else int.from_bytes(os.urandom(4), "little")
count = [0]


@seed(rseed)
@settings(max_examples=examples, deadline=None, derandomize=False, database=None, suppress_health_check=list(HealthCheck))
@given(problems(n_max=n_max, m_max=80))
def campaign(prob):
    count[0] += 1
    check_problem(*prob, grad=True)


with installed(sanitize="asan" if os.environ.get("FUZZ_ASAN") else False):
    print(f"seed {rseed}, {examples} examples, n_max {n_max}", flush=True)
    campaign()
    print(f"ok: {count[0]} problems agree with the oracle", flush=True)
