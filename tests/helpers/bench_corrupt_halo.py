"""bench.py with a halo transport that DAMAGES what arrives (one bit of the received pyramid): the fault
tests/test_gpu_bench_contract.py::test_bench_halo_check_failure_is_collective injects.  Launched under torch.distributed.run
in place of bench.py; the product file carries no test hook."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402

_make = bench._torch_halo_transport


def _corrupting(rank, world, staged):
    fn = _make(rank, world, staged)

    def wrapped(send_ptr, recv_ptr, nbytes, stream_ptr):
        fn(send_ptr, recv_ptr, nbytes, stream_ptr)
        if rank > 0:
            stream = torch.cuda.ExternalStream(stream_ptr)
            with torch.cuda.stream(stream):
                recv = torch.as_tensor(bench._DevMem(recv_ptr, nbytes), device="cuda")
                recv[12345] ^= 0x40
            stream.synchronize()
    return wrapped


bench._torch_halo_transport = _corrupting
bench.main()
