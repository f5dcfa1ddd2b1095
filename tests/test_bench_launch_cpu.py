"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run around it) — the argument / environment
plumbing, on CPU: every rank gets the variables torch.distributed.run would set, rank 0 owns stdout (the ONE JSON line), the
command line is passed through unchanged, a failing rank's exit code becomes the launcher's and takes the others down."""
import json
import os
import subprocess
import sys
import textwrap
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_rank_environments():
    envs = bench.rank_environments(4, 29511, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" == e["LOCAL_WORLD_SIZE"] for e in envs)
    assert all(e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" and e["PATH"] == "/bin" for e in envs)
    assert envs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"  # an explicit setting is respected ...
    assert bench.rank_environments(1, 1, base={})[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # ... the default is dmabuf IPC


def _launch(tmp_path, body, argv):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(body))
    code = "import sys; sys.path.insert(0, %r); import bench; raise SystemExit(bench.self_launch(3, %r, script=%r))" % (
        REPO, argv, str(script))
    return subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)


def test_self_launch_starts_all_ranks_and_rank0_owns_stdout(tmp_path):
    proc = _launch(tmp_path, """
        import json, os, sys
        rec = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(%r, "rank%%s.json" %% rec["RANK"]), "w").write(json.dumps(rec))
        print(json.dumps(rec))  # only rank 0's line may reach the launcher's stdout
        """ % str(tmp_path), ["--gpus", "3", "--steps", "7"])
    assert proc.returncode == 0, proc.stderr.decode()
    lines = proc.stdout.decode().strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["RANK"] == "0"
    recs = [json.load(open(str(tmp_path / ("rank%d.json" % r)))) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and all(r["WORLD_SIZE"] == "3" for r in recs)
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and all(r["argv"] == ["--gpus", "3", "--steps", "7"] for r in recs)


def test_self_launch_propagates_a_failing_rank(tmp_path):
    t0 = time.time()
    proc = _launch(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(11)
        time.sleep(60)  # the healthy ranks would hang in their next collective: the launcher must take them down
        """, [])
    assert proc.returncode == 11
    assert time.time() - t0 < 30


def test_gpus_must_match_the_launcher(tmp_path):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=120)
    assert proc.returncode != 0 and b"launcher started 1 rank" in proc.stderr
