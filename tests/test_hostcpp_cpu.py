"""CPU check of the C++17 host layer: the header and its driver compile warning-free as plain C++17 against the C ABI
header (no hipcc, no reference headers), and link against libls1hip.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_compiles_as_plain_cpp17(tmp_path):
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "ls1-mardyn_amd", "host"), os.path.join(ROOT, "tests", "hostcpp", "host_sim.cpp")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr


def test_host_driver_links_against_the_abi_library():
    lib = os.path.join(ROOT, "ls1-mardyn_amd", "lib", "libls1hip.so")
    assert os.path.exists(lib), "libls1hip.so not built"
    res = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostcpp")], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert os.path.exists(os.path.join(ROOT, "tests", "hostcpp", "host_sim"))
