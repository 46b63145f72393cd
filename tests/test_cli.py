"""The command line tool (decompress_amd/cli.py = bin/decompress.ml): argument handling here, the cram test of the
reference (test/bin/simple.t) step by step on the GPU with python's zlib / gzip in the place of zpipe.c."""
import gzip
import os
import subprocess
import sys
import zlib

import pytest

from decompress_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arguments_like_the_reference():
    """bin/decompress.ml:264-296: -d, -f/--format (case-insensitive, default deflate), -l/--level (default 4, >= 0),
    two optional positions"""
    assert cli.parse([]) == (False, "deflate", 4, None, None)
    assert cli.parse(["-d", "-fzlib"]) == (True, "zlib", 4, None, None)
    assert cli.parse(["-f", "GZip", "--level", "0", "a", "b"]) == (False, "gzip", 0, "a", "b")
    assert cli.parse(["--format=lzo", "-l9", "in"]) == (False, "lzo", 9, "in", None)
    for bad, msg in ((["-f", "bzip"], 'Invalid format: "bzip"'), (["-l", "-1"], "The compression level must be positive"),
                     (["-l", "x"], "Invalid level"), (["a", "b", "c"], "too many arguments")):
        with pytest.raises(ValueError, match=msg.replace('"', '.')):
            cli.parse(bad)
    assert cli.main(["-f", "bzip"]) == cli.CLI_ERROR


def _tool(args, data=b"", cwd=None):
    r = subprocess.run([sys.executable, "-m", "decompress_amd.cli"] + args, input=data, capture_output=True, cwd=cwd or ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT), timeout=300)
    return r.returncode, r.stdout, r.stderr.decode()


@pytest.mark.gpu
def test_simple_t(tmp_path, oracle):
    """test/bin/simple.t:1-28, with zlib.decompress / zlib.compress / gzip standing in for ./zpipe"""
    from decompress_amd import workloads
    hello = b"Hello World!\n"
    rc, z, _ = _tool(["-d"], hello)                      # echo "Hello World!" | decompress -d > simple.z
    assert rc == 0 and z == oracle.deflate_raw(hello, level=4, queue=4096, driver=oracle.DRV_CLI)[0]
    assert _tool([], z)[:2] == (0, hello)                # decompress < simple.z
    rc, z, _ = _tool(["-d", "-fzlib"], hello)
    assert rc == 0 and zlib.decompress(z) == hello       # ./zpipe -d < simple.z
    assert _tool(["-fzlib"], z)[:2] == (0, hello)
    corpus = workloads.corpus()
    news, bib = corpus["news"], corpus["bib"]
    (tmp_path / "news").write_bytes(news)
    (tmp_path / "bib").write_bytes(bib)
    assert _tool(["-fzlib"], zlib.compress(news))[:2] == (0, news)          # ./zpipe < news > news.z; decompress -fzlib
    rc, z, _ = _tool(["-fzlib", "-d"], news)
    assert rc == 0 and zlib.decompress(z) == news
    rc, z, _ = _tool(["-fgzip", "-d"], news)
    assert rc == 0 and gzip.decompress(z) == news and z[9] == 3            # OS = Unix
    assert _tool(["-fgzip"], z)[:2] == (0, news)
    for level in ([], ["--level", "0"]):                                     # file names instead of the standard streams
        assert _tool(["-fzlib", "-d"] + level + [str(tmp_path / "bib"), str(tmp_path / "bib.zlib")])[0] == 0
        assert _tool(["-fzlib", str(tmp_path / "bib.zlib"), str(tmp_path / "bib.out")])[0] == 0
        assert (tmp_path / "bib.out").read_bytes() == bib
    assert _tool(["-fgzip", "-d", "--level", "0", str(tmp_path / "news"), str(tmp_path / "news.gz")])[0] == 0
    assert gzip.decompress((tmp_path / "news.gz").read_bytes()) == news
    # LZO both ways, and a malformed stream: the reference's message with its full stop, cmdliner's status
    rc, z, _ = _tool(["-flzo", "-d"], bib)
    assert rc == 0 and _tool(["-flzo"], z)[:2] == (0, bib)
    rc, out, err = _tool(["-fzlib"], zlib.compress(news)[:5000])
    assert rc == cli.CLI_ERROR and err.strip().splitlines()[-1] == "decompress: Unexpected end of input." and news.startswith(out)
