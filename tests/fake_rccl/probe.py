"""TEST INFRASTRUCTURE.  The stand-in collective library on its own: W processes on ONE device, K all-gathers each, every received
block checked against what the sending rank put in (a function of rank, gather number and position).  Run as

    python tests/fake_rccl/probe.py [--world 2] [--transport ipc|shm] [--gathers 200] [--count 3200]

The parent never touches the GPU; the ranks are fresh interpreters (subprocess), the id travels on the command line."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfake_rccl.so")


def load():
    lib = C.CDLL(LIB, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    return lib


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def payload(torch, rank, n, count, device):
    i = torch.arange(count, dtype=torch.float32, device=device)
    return i * 0.001 + float(rank) * 1000.0 + float(n)


def rank_main(a):
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = load()
    uid = UniqueId()
    C.memmove(C.byref(uid), bytes.fromhex(a.id), 128)
    comm = C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.ncclCommInitRank(C.byref(comm), a.world, uid, a.rank)
    if rc != 0:
        print(json.dumps({"rank": a.rank, "ok": False, "where": "init", "error": lib.ncclGetErrorString(rc).decode()}), flush=True)
        return 1
    t_init = time.perf_counter() - t0
    st = torch.cuda.Stream(device=dev)
    recv = [torch.zeros(a.world, a.count, dtype=torch.float32, device=dev) for _ in range(2)]
    bad = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        for n in range(1, a.gathers + 1):
            send = payload(torch, a.rank, n, a.count, dev)
            rc = lib.ncclAllGather(send.data_ptr(), recv[n & 1].data_ptr(), a.count, 7, comm, st.cuda_stream)
            if rc != 0:
                print(json.dumps({"rank": a.rank, "ok": False, "where": f"gather {n}", "error": lib.ncclGetErrorString(rc).decode()}), flush=True)
                return 1
            send.record_stream(st)
            if n % 16 == 0 or n == a.gathers:          # check on the stream, without a host sync in between
                for r in range(a.world):
                    bad += int((recv[n & 1][r] != payload(torch, r, n, a.count, dev)).sum().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.ncclCommDestroy(comm)
    print(json.dumps({"rank": a.rank, "ok": bad == 0, "mismatches": bad, "gathers": a.gathers, "count": a.count, "init_s": round(t_init, 3),
                      "us_per_gather_incl_checks": round(1e6 * dt / a.gathers, 1), "transport": os.environ.get("FAKE_RCCL_TRANSPORT", "ipc")}), flush=True)
    return 0 if bad == 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--transport", default="ipc")
    ap.add_argument("--gathers", type=int, default=200)
    ap.add_argument("--count", type=int, default=3200)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--id", default="")
    a = ap.parse_args()
    if a.rank >= 0:
        sys.exit(rank_main(a))
    lib = load()
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    env = dict(os.environ, FAKE_RCCL_TRANSPORT=a.transport, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--world", str(a.world), "--gathers", str(a.gathers), "--count", str(a.count),
                               "--rank", str(r), "--id", bytes(uid)[:128].hex()], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(a.world)]
    rc = 0
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        print(out.strip() or json.dumps({"rank": r, "ok": False, "stderr": err[-800:]}))
        if p.returncode != 0:
            rc = 1
            print(err[-1500:], file=sys.stderr)
    sys.exit(rc)


if __name__ == "__main__":
    main()
