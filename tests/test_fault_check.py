"""scripts/ablate/check_fault.sh: the ablation / timing recipes must fail when the HIP runtime printed a GPU fault,
whatever the exit status of the pipeline that produced the log (round 2's call15.log exited 0 with a fault in it)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "ablate", "check_fault.sh")


def test_fault_text_makes_the_check_fail(tmp_path):
    bad = tmp_path / "time.err"
    bad.write_text("NOMEL\nMemory access fault by GPU node-2 (Agent handle: 0x5c587d053aa0) on address "
                   "0x71e1bfe00000. Reason: Unknown.\nclips=256 frames=1439232 ms=1.402\n")
    good = tmp_path / "ok.log"
    good.write_text("clips=256 frames=1439232 ms=1.182\n")
    assert subprocess.run(["bash", SCRIPT, str(good)]).returncode == 0
    r = subprocess.run(["bash", SCRIPT, str(good), str(bad)], capture_output=True, text=True)
    assert r.returncode == 3 and "GPU fault reported" in r.stderr
    assert subprocess.run(["bash", SCRIPT, str(tmp_path / "missing.log")]).returncode == 0
