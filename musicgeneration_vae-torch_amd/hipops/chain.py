"""Launch chains: the host side of mgvae_chain_run (include/mgvae.h, csrc/chain.hip).

A ``Chain`` is a prepared list of C-ABI calls -- the forward or the backward of one block -- whose arguments live in ONE
array of 8-byte words (call arguments first, then the structs some arguments point to).  What changes from step to step
is where the tensors are: every pointer argument names a *slot*, and ``run`` writes ``slot address + byte offset`` into
the words that need it (one vectorised numpy store) and makes ONE foreign call.  Everything else -- geometry, flags,
strides, workspace sizes -- is constant for a (block, shape, storage type) and was encoded when the chain was built.

The entry points, their argument order and their meaning are exactly those of the eager path (hipops/functional.py calls
them one by one); a chain only removes the per-launch trip through Python, ctypes and the autograd engine.
"""
import ctypes
import struct

import numpy as np

from . import _native as nat


class Slot:
    """a tensor address supplied at run time; ``slot + n`` = the same address n BYTES further"""
    __slots__ = ("index", "offset")

    def __init__(self, index, offset=0):
        self.index, self.offset = index, offset

    def __add__(self, nbytes):
        return Slot(self.index, self.offset + int(nbytes))


class _StructRef:
    __slots__ = ("word",)

    def __init__(self, word):
        self.word = word


_FN_IDS = {}


def _fn_id(name):
    i = _FN_IDS.get(name)
    if i is None:
        i = nat.lib().mgvae_chain_fn_id(name.encode())
        if i < 0:
            raise RuntimeError("%s is not an entry point the chain interpreter knows (tools/gen_chain_dispatch.py)" % name)
        _FN_IDS[name] = i
    return i


def _f32_bits(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _f64_bits(v):
    return struct.unpack("<Q", struct.pack("<d", float(v)))[0]


class Chain:
    def __init__(self):
        self._calls = []          # (fn id, nargs, first word)
        self._words = []          # python ints (uint64), call arguments in call order
        self._patch = []          # (word index, slot index, byte offset)
        self._structs = []        # (first word relative to the struct area, [words], [(rel word, Slot)])
        self._swords = 0
        self._struct_args = []    # (word index, struct-area word)
        self.nslots = 0
        self._final = False

    # ------------------------------------------------------------------ building
    def slot(self):
        s = Slot(self.nslots)
        self.nslots += 1
        return s

    def slots(self, n):
        return [self.slot() for _ in range(n)]

    def struct(self, obj, pointers=()):
        """keep a ctypes Structure inside the chain; ``pointers``: (field name, Slot) for pointer fields filled at run time"""
        raw = bytes(obj)
        raw += b"\0" * (-len(raw) % 8)
        words = list(struct.unpack("<%dQ" % (len(raw) // 8), raw))
        ptrs = []
        for field, sl in pointers:
            off = getattr(type(obj), field).offset
            if off % 8:
                raise ValueError("pointer field %s is not 8-byte aligned" % field)
            ptrs.append((off // 8, sl))
        ref = _StructRef(self._swords)
        self._structs.append((self._swords, words, ptrs))
        self._swords += len(words)
        return ref

    def call(self, name, *args):
        res, argtypes = nat.SIGNATURES[name]
        if res is not ctypes.c_int:
            raise ValueError("%s does not return a status code" % name)
        if len(args) != len(argtypes):
            raise ValueError("%s takes %d arguments, got %d" % (name, len(argtypes), len(args)))
        first = len(self._words)
        for a, t in zip(args, argtypes):
            w = 0
            if isinstance(a, Slot):
                self._patch.append((len(self._words), a.index, a.offset))
            elif isinstance(a, _StructRef):
                self._struct_args.append((len(self._words), a.word))
            elif a is None:
                w = 0
            elif t is ctypes.c_float:
                w = _f32_bits(a)
            elif t is ctypes.c_double:
                w = _f64_bits(a)
            elif t in (ctypes.c_int, ctypes.c_int32):
                w = int(a) & 0xFFFFFFFF
            elif t in (ctypes.c_size_t, ctypes.c_uint64, ctypes.c_long):
                w = int(a) & 0xFFFFFFFFFFFFFFFF
            else:                                   # a raw address that never changes (a persistent device table ...)
                w = int(a) & 0xFFFFFFFFFFFFFFFF
            self._words.append(w)
        self._calls.append((_fn_id(name), len(argtypes), first))

    def finalize(self):
        nargw = len(self._words)
        buf = np.zeros(nargw + self._swords, dtype=np.uint64)
        buf[:nargw] = np.array(self._words, dtype=np.uint64) if nargw else 0
        patch = list(self._patch)
        for first, words, ptrs in self._structs:
            buf[nargw + first:nargw + first + len(words)] = np.array(words, dtype=np.uint64)
            for rel, sl in ptrs:
                patch.append((nargw + first + rel, sl.index, sl.offset))
        base = buf.ctypes.data
        for widx, sword in self._struct_args:
            buf[widx] = base + 8 * (nargw + sword)
        self._buf = buf
        self._buf_p = ctypes.c_void_p(base)
        self._pidx = np.array([p[0] for p in patch], dtype=np.int64)
        self._pslot = np.array([p[1] for p in patch], dtype=np.int64)
        self._poff = np.array([p[2] for p in patch], dtype=np.uint64)
        self._carr = (nat.ChainCall * max(1, len(self._calls)))(*[nat.ChainCall(f, n, w, 0) for f, n, w in self._calls])
        self._ncalls = len(self._calls)
        self._failed = ctypes.c_int(-1)
        self._failed_ref = ctypes.byref(self._failed)
        self._run = nat.lib().mgvae_chain_run
        self._names = [None] * self._ncalls
        self._final = True
        return self

    # ------------------------------------------------------------------ running
    def run(self, addresses):
        """``addresses``: one integer per slot (tensor.data_ptr(), a raw stream handle, 0 for an absent tensor)"""
        if len(addresses) != self.nslots:
            raise ValueError("chain has %d slots, got %d addresses" % (self.nslots, len(addresses)))
        if len(self._pidx):
            self._buf[self._pidx] = np.array(addresses, dtype=np.uint64)[self._pslot] + self._poff
        rc = self._run(self._carr, self._ncalls, self._buf_p, self._failed_ref)
        if rc != 0:
            inv = {v: k for k, v in _FN_IDS.items()}
            which = inv.get(self._calls[self._failed.value][0], "?") if 0 <= self._failed.value < self._ncalls else "?"
            nat.check(rc, "chain call %d (%s)" % (self._failed.value, which))

    def __len__(self):
        return len(self._calls)
