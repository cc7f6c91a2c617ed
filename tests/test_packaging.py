"""`pip install .` packaging (the reference is consumed through `pip install` of its two thirdparties, /root/reference/README.md:24-28;
Edit_core/thirdparties/diff-gaussian-rasterization/setup.py:17-34): a wheel built from this tree holds the three packages and the native
libraries, and -- unpacked somewhere else, with nothing of this repository on the path -- provides the reference's import names."""
import glob
import os
import subprocess
import sys
import zipfile

from tests import util


def test_wheel_provides_the_reference_import_names(tmp_path):
    out = tmp_path / "whl"
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", util.ROOT, "--no-build-isolation", "--no-deps", "-q", "-w", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    whl = glob.glob(str(out / "youreditableavatar_amd-*.whl"))
    assert len(whl) == 1 and "none-any" not in os.path.basename(whl[0])            # carries compiled code: platform-tagged
    z = zipfile.ZipFile(whl[0])
    site = tmp_path / "site"
    names = set()
    for m in z.namelist():                            # what `pip install` does with it: <dist>.data/purelib/ and the root both land in site-packages
        rel = m.split(".data/purelib/", 1)[-1].split(".data/platlib/", 1)[-1]
        names.add(rel)
        if not m.endswith("/"):
            dst = site / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            dst.write_bytes(z.read(m))
    for must in ("diff_gaussian_rasterization/__init__.py", "simple_knn/__init__.py", "simple_knn/_C.py", "youreditableavatar_amd/lib/libtgs_raster.so",
                 "youreditableavatar_amd/lib/.source_hash", "youreditableavatar_amd/diff_gaussian_rasterization/_C.py", "youreditableavatar_amd/simple_knn/_C.py"):
        assert must in names, must
    assert any(n.startswith("youreditableavatar_amd/diff_gaussian_rasterization/_Cext") and n.endswith(".so") for n in names)
    assert not any(n.startswith(("oracle/", "tests/")) for n in names)              # the checker is not part of the product
    code = ("import diff_gaussian_rasterization as d, simple_knn, simple_knn._C as k, youreditableavatar_amd as y, os;"
            "assert os.path.dirname(y.__file__).startswith(%r), y.__file__;"
            "print(d.GaussianRasterizationSettings._fields[:2], d.GaussianRasterizer.__name__, callable(k.distCUDA2), d._C._lib.tgs_abi_version())" % str(site))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(site)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "('image_height', 'image_width') GaussianRasterizer True 3" in r.stdout
    for d in ("build", "youreditableavatar_amd.egg-info"):                            # pip's in-tree leftovers
        subprocess.run(["rm", "-rf", os.path.join(util.ROOT, d)])
