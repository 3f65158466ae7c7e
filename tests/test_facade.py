"""The drop-in C++ headers (include/orb_slam/...) compile against the C ABI without OpenCV/ROS, read the
reference's parameter files, and (GPU) run extract -> match -> F/H through the facade classes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "facade_check")


def _build():
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++17", "-DORBX_NO_OPENCV", "-DORBX_NO_ROS", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "facade_check.cpp"), "-o", EXE, "-L" + os.path.join(ROOT, "orb_slam_amd"),
           "-lorbx", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "orb_slam_amd"), "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.check_call(cmd)


def test_facade_compiles_and_reads_reference_yaml():
    _build()
    out = subprocess.check_output([EXE, "norun", os.path.join(ROOT, "tests", "golden", "cfg")], text=True)
    assert "n_key_points=1000" in out


def test_shared_facade_objects_are_thread_safe_under_tsan():
    """One ORBMatcher and one ORBExtractor shared by two threads, as the reference's static members are shared by
    the tracker and the local mapper: clean under ThreadSanitizer (host code only; the C ABI is a stand-in whose
    handles hold plain state, so sharing a handle between threads would be reported), and the sanitizer does object
    when the same calls go to ONE raw handle from two threads."""
    exe = os.path.join(ROOT, "tests", "facade_threads")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-DORBX_NO_OPENCV", "-DORBX_NO_ROS",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "facade_threads.cpp"),
                           "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("LD_PRELOAD", None)
    ok = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "ThreadSanitizer" not in ok.stderr, ok.stderr[-2000:]
    assert "matchers=3 extractors=3" in ok.stdout
    racy = subprocess.run([exe, "shared"], env=env, capture_output=True, text=True, timeout=120)
    assert racy.returncode == 66 and "data race" in racy.stderr


@pytest.mark.gpu
def test_facade_runs_on_gpu():
    if not os.path.exists(EXE):
        _build()
    out = subprocess.check_output([EXE, "run", os.path.join(ROOT, "tests", "golden", "cfg")], text=True, timeout=120)
    assert "levels=8" in out and "H empty=0" in out
    kp = int(out.split("kp=")[1].split()[0])
    assert kp > 300 and f"desc={kp}" in out and f"matches=" in out
    assert "sf1=1.20000005" in out
    assert int(out.split("bow_matches=")[1].split()[0]) > kp // 2
    assert int(out.split("point_matches=")[1].split()[0]) > kp // 2
    assert int(out.split("epipolar_matches=")[1].split()[0]) > kp // 5
    fused = int(out.split("fused=")[1].split()[0])
    attached = int(out.split("attached=")[1].split()[0])
    assert fused > kp // 2 and 0 < attached <= fused
    assert int(out.split("fused_again=")[1].split()[0]) == 0
    pose = out.split("pose ok=")[1]
    assert pose.startswith("1 tri=1")
    inl = int(pose.split("inliers=")[1].split()[0])
    assert inl > 250 and int(pose.split("in_front=")[1].split()[0]) == inl
    assert abs(float(pose.split("r00=")[1].split()[0]) - 0.9988) < 2e-3   # cos(0.05)
    assert float(pose.split("tx=")[1].split()[0]) > 0.9                   # t = (0.3, 0.02, 0.05) normalised
    assert "two_threads_same=1" in out and "seam_same_F=1" in out
