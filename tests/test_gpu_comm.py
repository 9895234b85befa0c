"""The library's own RCCL binding (wd_comm_* / wd_allreduce_counts), single rank: RCCL is
dlopen'ed, a communicator of world size 1 is created and an int64 sum all-reduce runs on the
context's stream.  (Multi-rank behaviour is a property of RCCL; the sharding and merge logic is
covered on CPU by tests/test_dist.py.)"""
import numpy as np
import pytest

from well_duplicates_amd.scanner import Scanner

pytestmark = pytest.mark.gpu


def test_single_rank_allreduce():
    with Scanner(0) as sc:
        uid = sc.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        sc.comm_init(0, 1, uid)
        host = np.arange(-50, 5000, dtype=np.int64) * 1234567891
        buf = sc.malloc(host.nbytes)
        sc.h2d(buf, host)
        sc.allreduce_counts(buf, host.shape[0])
        sc.synchronize()
        assert (sc.d2h(buf, host.nbytes, np.int64) == host).all()     # sum over one rank
        sc.allreduce_counts(buf, 0)
        sc.comm_destroy()
        with pytest.raises(RuntimeError):                              # no communicator any more
            sc.allreduce_counts(buf, 4)
        sc.free(buf)
