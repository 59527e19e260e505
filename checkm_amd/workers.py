"""Worker processes of MarkerGeneFinder.find on a multi-GPU node: one per device, spawned by find() itself.

The reference's find() forks its own bin workers (checkm/markerGeneFinder.py:59-83: a process per thread, bins handed out through a
queue), so a CheckM command is ONE process whatever the parallelism.  The same holds here: on a node with several MI355X the command
stays one process that runs everything outside the marker-gene path once (tree placement, bin statistics, the storage/ files, the
printing), and find() starts one worker per GPU.  A worker owns its device contexts, scans a size-balanced shard of the bins
(dist.shard_bins), writes the tables of the bins it owns and keeps their packed hits resident; `analyseResults` / `printSummary` of the
parent ask the workers to reduce their bins, the QA rows travel by ONE all_gather among the workers (RCCL over xGMI; gloo when two
workers share a device, as in the tests) and rank 0 hands the table to the parent.  Everything that needs the hits themselves in the
parent (output formats 3-9, cacheResults, the aligner) reads the tables the workers wrote, as a later `checkm qa` would.

`torchrun`-style launches (WORLD_SIZE > 1 in the environment: bench.py --gpus N) keep working as before: every rank is then a full
process and no workers are spawned.  The fan-out is OPT-IN: CKM_GPUS selects the devices ("all" = every visible device; "0,2,5"); unset,
empty or a single device = scan in this process on one GPU (a find() that quietly spawns a process per GPU is not what a caller of
the reference's API expects, and no multi-GPU hardware run of this path exists yet: the driver measures scaling through torchrun).
A worker error tears the whole pool down (peers may sit in a collective); CKM_WORKER_TIMEOUT_S bounds the time WITHOUT any reply."""
import atexit
import multiprocessing as mp
import os
import socket
import sys
import traceback

_POOL = None


def devices():
    """The devices find() fans out over, or None to scan in this process."""
    if os.environ.get("CKM_WORKER") or os.environ.get("CKM_EMULATE_RANK") or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return None
    spec = os.environ.get("CKM_GPUS", "").strip()
    if spec == "all":
        from checkm_amd import _lib
        devs = list(range(_lib.device_count()))
    else:
        devs = [int(x) for x in spec.split(",") if x.strip() != ""]
    if len(devs) < 2:
        return None
    # never from a process that is itself being spawned: a caller whose script has no `if __name__ == "__main__"` guard re-runs its top
    # level in every 'spawn' child, and a find() reached that way must scan in place instead of starting workers of its own
    cur = mp.current_process()
    if cur.name != "MainProcess" or getattr(cur, "_inheriting", False):
        return None
    main_file = getattr(sys.modules.get("__main__"), "__file__", None)
    if main_file is not None and not os.path.exists(main_file):
        return None                           # 'spawn' re-imports __main__ by path: a script read from stdin cannot have workers
    return devs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class WorkerError(RuntimeError):
    pass


class Pool(object):
    def __init__(self, devs):
        from checkm_amd.defaultValues import DefaultValues
        ctx = mp.get_context("spawn")
        self.devs = list(devs)
        port = _free_port()
        backend = os.environ.get("CKM_DIST_BACKEND") or ("gloo" if len(set(devs)) < len(devs) else "nccl")
        defaults = {k: getattr(DefaultValues, k) for k in dir(DefaultValues) if k.isupper() and isinstance(getattr(DefaultValues, k), (str, int, float))}
        self.conns, self.procs = [], []
        # CKM_WORKER travels in the environment the children are spawned with (they re-import __main__ BEFORE _worker_main runs)
        had = os.environ.get("CKM_WORKER")
        os.environ["CKM_WORKER"] = "1"
        try:
            for r, d in enumerate(self.devs):
                pc, cc = ctx.Pipe()
                p = ctx.Process(target=_worker_main, args=(r, len(self.devs), d, cc, port, backend, defaults, list(sys.path)), daemon=True)
                p.start()
                cc.close()
                self.conns.append(pc); self.procs.append(p)
        finally:
            if had is None:
                os.environ.pop("CKM_WORKER", None)
            else:
                os.environ["CKM_WORKER"] = had
        self._collect()                          # every worker reports in (library loaded, device usable) before any work is sent

    def _collect(self):
        """One reply from every worker, in whatever order they arrive.  A worker that reports an error or dies ends the WHOLE pool at once:
        its peers may be blocked in a collective that will never complete (they are terminated), and the parent must not wait on them."""
        import time
        out, pending = [None] * len(self.conns), set(range(len(self.conns)))
        limit = float(os.environ.get("CKM_WORKER_TIMEOUT_S", "3600"))
        t0 = time.monotonic()
        while pending:
            progressed = False
            for r in sorted(pending):
                conn, proc = self.conns[r], self.procs[r]
                try:
                    if conn.poll(0):
                        kind, payload = conn.recv()
                        if kind != "ok":
                            self.abort()
                            raise WorkerError(payload)
                        out[r] = payload; pending.discard(r); progressed = True
                        t0 = time.monotonic()              # (the limit is on silence, not on the length of a call: a long find over many bins keeps answering)
                    elif not proc.is_alive():
                        self.abort()
                        raise WorkerError("GPU worker %d (device %d) died (exit code %s)" % (r, self.devs[r], proc.exitcode))
                except (EOFError, OSError):
                    self.abort()
                    raise WorkerError("GPU worker %d (device %d) died" % (r, self.devs[r]))
            if pending and not progressed:
                if time.monotonic() - t0 > limit:
                    self.abort()
                    raise WorkerError("GPU workers did not answer within %.0f s (CKM_WORKER_TIMEOUT_S)" % limit)
                time.sleep(0.002)
        return out

    def call(self, cmd, args=None, per_worker=None):
        """Send `cmd` to every worker (args: dict for all; per_worker: list of dicts merged into it), wait for all, return the replies."""
        for r, c in enumerate(self.conns):
            a = dict(args or {})
            if per_worker is not None:
                a.update(per_worker[r])
            c.send((cmd, a))
        return self._collect()

    def abort(self):
        """Terminate every worker (one failed: the others may sit in a collective) and forget the pool."""
        global _POOL
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        for p in self.procs:
            p.join(timeout=10)
        for c in self.conns:
            try:
                c.close()
            except Exception:
                pass
        self.conns, self.procs = [], []
        if _POOL is self:
            _POOL = None

    def close(self):
        for c in self.conns:
            try:
                c.send(("close", {}))
            except Exception:
                pass
        for p in self.procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
        self.conns, self.procs = [], []


def get_pool(devs):
    global _POOL
    if _POOL is not None and _POOL.devs != list(devs):
        _POOL.close()
        _POOL = None
    if _POOL is None:
        _POOL = Pool(devs)
        atexit.register(shutdown)
    return _POOL


def shutdown():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL = None


class _SelectedOnly(object):
    """What the workers need of a bin's BinMarkerSets: its selected marker set."""

    def __init__(self, ms):
        self._ms = ms

    def selectedMarkerSet(self):
        return self._ms


class _Het(object):
    def __init__(self, het):
        self.aaiMeanBinHetero = het


def _worker_main(rank, world, device, conn, port, backend, defaults, path):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), CHECKM_AMD_DEVICE=str(device), CKM_WORKER="1",
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CKM_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("CKM_GPUS", None)
    for p in path:
        if p not in sys.path:
            sys.path.append(p)
    try:
        import logging
        logging.getLogger('timestamp').addHandler(logging.StreamHandler(sys.stderr))
        logging.getLogger('timestamp').setLevel(logging.WARNING)
        from checkm_amd import dist as cdist
        from checkm_amd import markerGeneFinder as mgf
        from checkm_amd import runtime
        from checkm_amd.defaultValues import DefaultValues
        from checkm_amd.resultsParser import ResultsParser
        for k, v in defaults.items():
            setattr(DefaultValues, k, v)
        runtime.get_ctx()                      # fails here, loudly, if the device is not a usable gfx950
        conn.send(("ok", None))
    except BaseException:
        conn.send(("error", "GPU worker %d (device %d) could not start:\n%s" % (rank, device, traceback.format_exc())))
        return
    models, parsers = {}, {}
    while True:
        try:
            cmd, a = conn.recv()
        except EOFError:
            break
        try:
            if cmd == "close":
                conn.send(("ok", None))
                break
            if cmd == "find":
                finder = mgf.MarkerGeneFinder(a["threads"])
                m = finder.find(a["binFiles"], a["outDir"], a["tableOut"], a["hmmerOut"], a["markerFile"], a["bKeepAlignment"], a["bNucORFs"], a["bCalledGenes"])
                key = (os.path.abspath(a["outDir"]), a["tableOut"])
                models[key] = m
                ent = mgf.SCAN_CACHE[key]
                conn.send(("ok", dict(heads=ent["profiles"].headers if rank == 0 else None, totals=ent["totals"], owned=sorted(ent["owned"]))))
            elif cmd == "analyse":
                key = (os.path.abspath(a["outDir"]), a["hmmTableFile"])
                rp = ResultsParser(models[key])
                rp.analyseResults(a["outDir"], a["binStatsFile"], a["hmmTableFile"], a["bIgnoreThresholds"], a["evalueThreshold"], a["lengthThreshold"],
                                  a["bSkipPseudoGeneCorrection"], a["bSkipAdjCorrection"])
                parsers[key] = rp
                conn.send(("ok", None))
            elif cmd == "summary":
                rp = parsers[(os.path.abspath(a["outDir"]), a["hmmTableFile"])]
                sets = {b: _SelectedOnly(ms) for b, ms in a["sets"].items()}
                table = rp._gather_rows(_Het(a["het"]), sets, a["bIndividualMarkers"], a["order"])     # ONE all_gather among the workers
                conn.send(("ok", table if rank == 0 else None))
            elif cmd == "release":
                for key in [k for k in parsers if a["outDir"] is None or k[0] == os.path.abspath(a["outDir"])]:
                    del parsers[key]
                for key in [k for k in models if a["outDir"] is None or k[0] == os.path.abspath(a["outDir"])]:
                    del models[key]
                mgf.release_scan(a["outDir"])
                conn.send(("ok", None))
            else:
                conn.send(("error", "unknown worker command %r" % (cmd,)))
        except SystemExit as e:                 # the product's error path is logger.error + sys.exit
            conn.send(("error", "GPU worker %d: the scan ended with exit status %s (see its messages above)" % (rank, e.code)))
        except BaseException:
            conn.send(("error", "GPU worker %d:\n%s" % (rank, traceback.format_exc())))
    try:
        cdist.shutdown()
        runtime.close()
    except Exception:
        pass
