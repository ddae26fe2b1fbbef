"""Lining up the ranks of a multi-GPU run — and nothing else.

The hot path has NO data-path collective (voices share nothing: MLDSPFunctional.h:321-349,
source/app/MLSynth.h:49-57), so all a multi-GPU job needs from its ranks is (1) a barrier on both
sides of the timed region, (2) the slowest rank's time, (3) a few facts per rank (which device it
ran on). That does not need RCCL or torch. Three interchangeable rendezvous objects:

  FileRendezvous    ranks are processes of one node (bench.py's own launcher, or any launcher that
                    exports RANK / WORLD_SIZE / MLGPU_RDV_DIR): files in a fresh directory
  ThreadRendezvous  ranks are host threads of ONE process, one engine + stream per device
                    (SURVEY §8e "one host thread + stream per device"): threading.Barrier
  GlooRendezvous    ranks were started by torch.distributed.run (the driver's command form):
                    torch.distributed with the gloo backend on CPU tensors — its store is already
                    there, and nothing touches the GPUs' interconnect
  SoloRendezvous    world size 1

All offer barrier(), gather(obj) -> list (on every rank, JSON-serialisable objects), max(x).
"""
import json
import os
import threading
import time


class SoloRendezvous:
    rank, world, kind = 0, 1, "single process"

    def barrier(self):
        pass

    def gather(self, obj):
        return [obj]

    def max(self, x):
        return float(x)

    def close(self):
        pass


class FileRendezvous:
    """Barrier and gather through files in `directory` (created fresh by the launcher, so no stale state).
    Every collective call has a sequence number; rank r publishes `<seq>.<r>` atomically (write + rename)
    and waits until all `world` files of that sequence exist. Polling interval 50 us: the skew it adds to a
    barrier is far below the >= 100 ms timed regions it brackets."""
    kind = "processes (file rendezvous)"

    def __init__(self, directory, rank, world, timeout_s=900.0):
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), float(timeout_s)
        self.seq = 0
        if not os.path.isdir(directory):
            raise RuntimeError(f"rendezvous directory {directory} does not exist")

    def _exchange(self, payload):
        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}.{self.rank}")
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write(payload)
        os.rename(tmp, mine)
        deadline = time.monotonic() + self.timeout
        paths = [os.path.join(self.dir, f"{self.seq}.{r}") for r in range(self.world)]
        pending = list(paths)
        while pending:
            pending = [p for p in pending if not os.path.exists(p)]
            if pending:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rendezvous {self.seq} timed out waiting for {pending}")
                if os.path.exists(os.path.join(self.dir, "abort")):
                    raise RuntimeError(f"rank {self.rank}: another rank aborted the run")
                time.sleep(50e-6)
        return paths

    def barrier(self):
        self._exchange("")

    def gather(self, obj):
        out = []
        for p in self._exchange(json.dumps(obj)):
            with open(p) as f:
                out.append(json.loads(f.read()))
        return out

    def max(self, x):
        return max(float(v) for v in self.gather(float(x)))

    def abort(self):
        try:
            open(os.path.join(self.dir, "abort"), "w").close()
        except OSError:
            pass

    def close(self):
        pass


class ThreadRendezvous:
    """Ranks are threads of this process. Make one with `ThreadRendezvous.group(world)`, hand element r to thread r."""
    kind = "threads (one process, one engine per device)"

    class _Shared:
        def __init__(self, world):
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank, world):
        self.shared, self.rank, self.world = shared, rank, world

    @classmethod
    def group(cls, world):
        sh = cls._Shared(world)
        return [cls(sh, r, world) for r in range(world)]

    def barrier(self):
        self.shared.barrier.wait()

    def gather(self, obj):
        self.shared.slots[self.rank] = json.loads(json.dumps(obj))
        self.shared.barrier.wait()
        out = list(self.shared.slots)
        self.shared.barrier.wait()      # nobody overwrites a slot before everyone has read
        return out

    def max(self, x):
        return max(float(v) for v in self.gather(float(x)))

    def abort(self):
        self.shared.barrier.abort()

    def close(self):
        pass


class GlooRendezvous:
    """Ranks started by `python -m torch.distributed.run`: MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE are in the
    environment and the launcher's store is already listening, so torch.distributed (gloo, CPU) is the cheapest way in."""
    kind = "torch.distributed.run (gloo rendezvous, no RCCL: the path has no collective)"

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        self.dist, self.rank, self.world = dist, int(rank), int(world)
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self._mine = True

    def barrier(self):
        self.dist.barrier()

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def max(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def run_rank_processes(argv, world, env_extra=None, grace_s=10.0, poll_s=0.05):
    """Start `world` rank processes of one node (rank r = `argv` with RANK / LOCAL_RANK / WORLD_SIZE / MLGPU_RDV_DIR exported),
    lined up through a fresh rendezvous directory, and WATCH them: the moment any rank ends with a non-zero status - an
    exception, a refusal, a segfault that never got to write anything - the `abort` file is written so that the ranks waiting in
    a FileRendezvous leave at once instead of timing out, and what is still running after `grace_s` is killed (these exact
    children, by pid). Returns (exit codes, rank 0's stdout)."""
    import shutil
    import subprocess
    import tempfile
    rdv_dir = tempfile.mkdtemp(prefix="mlgpu_rdv_")
    procs = []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MLGPU_RDV_DIR=rdv_dir, **(env_extra or {}))
            procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out0 = []
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)  # never let rank 0 block on a full pipe
        reader.start()
        failed_at = None
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            if failed_at is None and any(rc not in (None, 0) for rc in rcs):
                failed_at = time.monotonic()
                try:
                    open(os.path.join(rdv_dir, "abort"), "w").close()
                except OSError:
                    pass
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            time.sleep(poll_s)
        reader.join(5.0)
        return [p.returncode for p in procs], "".join(out0)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv_dir, ignore_errors=True)


def from_environment():
    """The rendezvous of a rank process, from what its launcher exported."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return SoloRendezvous()
    if os.environ.get("MLGPU_RDV_DIR"):
        return FileRendezvous(os.environ["MLGPU_RDV_DIR"], rank, world)
    return GlooRendezvous(rank, world)
