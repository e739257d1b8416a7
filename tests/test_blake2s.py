"""BLAKE2s-256 of byte-string batches (include/zkw.h: zkw_blake2s256 / zkw_blake2s256_device).

The reference only re-exports the `blake2` crate (/root/reference/src/lib.rs:21) and holds no vector for it, so the pin
is the function's own specification: the test vector of RFC 7693 Appendix B ("abc"), the digest of the empty string
quoted in SURVEY Appendix C, and `hashlib.blake2s` (an independent implementation) over every length around the block
boundaries.  The oracle (oracle/hashes.hpp, written from the RFC's pseudo-code) is checked against those first, then the
product — compiled by g++ against the single-lane HIP stand-in here, and the real kernel on the GPU."""
import ctypes as C
import hashlib
import random

import numpy as np
import pytest

from era_zk_evm_amd import capi as K

RFC7693_ABC = bytes.fromhex("508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982")
EMPTY = bytes.fromhex("69217a3079908094e11121d042354a7c1f55b6482ca1a51e1b250dfd1ed0eef9")


def messages(seed, n_random=40):
    rng = random.Random(seed)
    out = [b"", b"abc"]
    for n in list(range(0, 70)) + [127, 128, 129, 191, 192, 193, 255, 256, 257, 1000, 4096, 4097]:
        out.append(bytes(rng.getrandbits(8) for _ in range(n)))
    for _ in range(n_random):
        out.append(bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 700))))
    out += [b"\x00" * 64, b"\xff" * 64, b"\xff" * 65, b"\x00" * 1]
    return out


def want(msgs):
    return [hashlib.blake2s(m).digest() for m in msgs]


def test_known_answers_of_the_specification():
    assert hashlib.blake2s(b"abc").digest() == RFC7693_ABC and hashlib.blake2s(b"").digest() == EMPTY


def test_oracle_matches_rfc_vector_and_hashlib(oracle):
    assert oracle.blake2s256([b"abc", b""]) == [RFC7693_ABC, EMPTY]
    msgs = messages(1)
    assert oracle.blake2s256(msgs) == want(msgs)


@pytest.fixture(scope="module")
def emu(isa):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import build_emu
    be = K.Backend(build_emu.build(), "zkw_").open(isa)
    yield be
    be.close()


def test_product_source_single_lane_matches_hashlib(emu, oracle):
    assert emu.blake2s256([b"abc", b""]) == [RFC7693_ABC, EMPTY]
    msgs = messages(2)
    got = emu.blake2s256(msgs)
    assert got == want(msgs) and got == oracle.blake2s256(msgs)
    assert emu.blake2s256([]) == []
    # ragged offsets: every message starts at a different byte misalignment, and a batch of empty messages only
    assert emu.blake2s256([b""] * 5) == [EMPTY] * 5


def test_argument_errors(emu):
    offs = np.array([0, 5, 3], dtype=np.uint64)
    data = (C.c_uint8 * 8)()
    out = (C.c_uint8 * 64)()
    with pytest.raises(K.ZkwError):
        emu.call("blake2s256", emu.ctx, data, offs.ctypes.data_as(C.c_void_p), C.c_uint32(2), out)
    with pytest.raises(K.ZkwError):
        emu.call("blake2s256", emu.ctx, data, None, C.c_uint32(2), out)


@pytest.fixture(scope="module")
def product(isa):
    be = K.load_product().open(isa)
    yield be
    be.close()


@pytest.mark.gpu
def test_gpu_matches_hashlib_and_oracle(product, oracle):
    assert product.blake2s256([b"abc", b""]) == [RFC7693_ABC, EMPTY]
    msgs = messages(3, n_random=400)
    got = product.blake2s256(msgs)
    assert got == want(msgs) and got == oracle.blake2s256(msgs)
    assert product.blake2s256([]) == [] and product.blake2s256([b""] * 130) == [EMPTY] * 130
    # one long message beside short ones (the lanes of its wave leave the block loop 16,000 iterations earlier), ragged start
    rng = random.Random(9)
    big = bytes(rng.getrandbits(8) for _ in range(1 << 20))
    batch = [b"x", big[: (1 << 20) - 3], b"", big[5:70000], b"tail"]
    assert product.blake2s256(batch) == want(batch)


@pytest.mark.gpu
def test_gpu_large_batch_sampled_and_device_entry(product):
    """65,536 messages of 0..300 bytes through the host entry (sampled against hashlib), then the same buffers through
    zkw_blake2s256_device on a torch stream: identical digests."""
    import torch
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 301, size=65536)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens, dtype=np.uint64)
    data = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    out = np.zeros((len(lens), 32), dtype=np.uint8)
    product.call("blake2s256", product.ctx, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), C.c_uint32(len(lens)), out.ctypes.data_as(C.c_void_p))
    raw = data.tobytes()
    for i in list(range(0, 65536, 257)) + [65535]:
        assert out[i].tobytes() == hashlib.blake2s(raw[int(offs[i]):int(offs[i + 1])]).digest(), i
    d_data = torch.from_numpy(np.concatenate([data, np.zeros(8, dtype=np.uint8)])).cuda()
    d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
    d_out = torch.zeros((len(lens), 32), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    product.call("blake2s256_device", product.ctx, C.c_void_p(d_data.data_ptr()), C.c_uint64(int(offs[-1])), C.c_void_p(d_offs.data_ptr()),
                 C.c_uint32(len(lens)), C.c_void_p(d_out.data_ptr()), C.c_void_p(stream.cuda_stream))
    stream.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), out)
