"""Host-side mirror of the kernel grammar of AutoGP.jl's GP module (src/GP.jl:39-517).

Same names, same constructor argument order and defaults as the reference structs so that
tests read like the reference's own (test/test_GP.jl:24-33).  The classes carry no
arithmetic: evaluating a covariance always goes through the HIP engine (engine.py); there
is no CPU path in this package.

Reference structs mirrored here (struct field order == parameter order in the program):
  WhiteNoise{value}                              src/GP.jl:131-133
  Constant{value}                                src/GP.jl:157-159
  Linear{intercept, bias=1, amplitude=1}         src/GP.jl:185-192
  SquaredExponential{lengthscale, amplitude=1}   src/GP.jl:228-234
  GammaExponential{lengthscale, gamma, amplitude=1}  src/GP.jl:269-277 (asserts 0 < gamma <= 2)
  Periodic{lengthscale, period, amplitude=1}     src/GP.jl:315-322
  Plus / Times {left, right, size, depth}        src/GP.jl:358-369, 404-415
  ChangePoint{left, right, location, scale, size, depth}  src/GP.jl:466-479
"""
from __future__ import annotations

import numpy as np

# opcodes = GPConfig codes (src/GP.jl:1101-1108), 0 = WhiteNoise (include/autogp_hip.h)
OP_WN, OP_CONST, OP_LIN, OP_SE, OP_GE, OP_PER, OP_PLUS, OP_TIMES, OP_CP = range(9)


class Node:
    """Abstract kernel expression (src/GP.jl:39-51)."""

    op: int = -1

    def __add__(self, other):   # Base.:+  (src/GP.jl:379)
        return Plus(self, other)

    def __mul__(self, other):   # Base.:*  (src/GP.jl:424)
        return Times(self, other)

    def size(self) -> int:
        raise NotImplementedError

    def depth(self) -> int:
        raise NotImplementedError

    def params(self):
        return ()

    def to_tuple(self):
        """Nested-tuple form understood by oracle/oracle.py (test infrastructure)."""
        raise NotImplementedError

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(p) for p in self._repr_args())})"

    def __eq__(self, other):
        return type(self) is type(other) and self._repr_args() == other._repr_args()

    def __hash__(self):
        return hash((type(self).__name__,) + tuple(self._repr_args()))


class LeafNode(Node):
    def size(self):
        return 1

    def depth(self):
        return 1

    def _repr_args(self):
        return tuple(self.params())


class BinaryOpNode(Node):
    def __init__(self, left: Node, right: Node):
        if not isinstance(left, Node) or not isinstance(right, Node):
            raise TypeError("operands must be kernel Nodes")
        self.left, self.right = left, right
        self._size = 1 + left.size() + right.size()
        self._depth = 1 + max(left.depth(), right.depth())

    def size(self):
        return self._size

    def depth(self):
        return self._depth

    def _repr_args(self):
        return (self.left, self.right) + tuple(self.params())


class WhiteNoise(LeafNode):
    op = OP_WN

    def __init__(self, value):
        self.value = value

    def params(self):
        return (self.value,)

    def to_tuple(self):
        return ("WN", float(self.value))


class Constant(LeafNode):
    op = OP_CONST

    def __init__(self, value):
        self.value = value

    def params(self):
        return (self.value,)

    def to_tuple(self):
        return ("C", float(self.value))


class Linear(LeafNode):
    op = OP_LIN

    def __init__(self, intercept, bias=1, amplitude=1):
        self.intercept, self.bias, self.amplitude = intercept, bias, amplitude

    def params(self):
        return (self.intercept, self.bias, self.amplitude)

    def to_tuple(self):
        return ("LIN", float(self.intercept), float(self.bias), float(self.amplitude))


class SquaredExponential(LeafNode):
    op = OP_SE

    def __init__(self, lengthscale, amplitude=1):
        self.lengthscale, self.amplitude = lengthscale, amplitude

    def params(self):
        return (self.lengthscale, self.amplitude)

    def to_tuple(self):
        return ("SE", float(self.lengthscale), float(self.amplitude))


class GammaExponential(LeafNode):
    op = OP_GE

    def __init__(self, lengthscale, gamma, amplitude=1):
        assert 0 < gamma <= 2            # src/GP.jl:274
        self.lengthscale, self.gamma, self.amplitude = lengthscale, gamma, amplitude

    def params(self):
        return (self.lengthscale, self.gamma, self.amplitude)

    def to_tuple(self):
        return ("GE", float(self.lengthscale), float(self.gamma), float(self.amplitude))


class Periodic(LeafNode):
    op = OP_PER

    def __init__(self, lengthscale, period, amplitude=1):
        self.lengthscale, self.period, self.amplitude = lengthscale, period, amplitude

    def params(self):
        return (self.lengthscale, self.period, self.amplitude)

    def to_tuple(self):
        return ("PER", float(self.lengthscale), float(self.period), float(self.amplitude))


class Plus(BinaryOpNode):
    op = OP_PLUS

    def to_tuple(self):
        return ("+", self.left.to_tuple(), self.right.to_tuple())


class Times(BinaryOpNode):
    op = OP_TIMES

    def to_tuple(self):
        return ("*", self.left.to_tuple(), self.right.to_tuple())


class ChangePoint(BinaryOpNode):
    op = OP_CP

    def __init__(self, left, right, location, scale):
        super().__init__(left, right)
        self.location, self.scale = location, scale

    def params(self):
        return (self.location, self.scale)

    def to_tuple(self):
        return ("CP", self.left.to_tuple(), self.right.to_tuple(), float(self.location), float(self.scale))


def unroll(node: Node):
    """Flat list of all sub-kernels in (left, right, node) order — src/GP.jl:111-113."""
    if isinstance(node, LeafNode):
        return [node]
    return unroll(node.left) + unroll(node.right) + [node]


def encode(node: Node):
    """Postfix program of the C ABI: (ops uint8[S], prm float64[...]) in `unroll` order."""
    seq = unroll(node)
    ops = np.fromiter((nd.op for nd in seq), dtype=np.uint8, count=len(seq))
    prm = np.array([float(v) for nd in seq for v in nd.params()], dtype=np.float64)
    return ops, prm


def encode_batch(nodes):
    """CSR-packed programs for a particle batch: (op_off, ops, prm_off, prm)."""
    op_off = np.zeros(len(nodes) + 1, dtype=np.int32)
    prm_off = np.zeros(len(nodes) + 1, dtype=np.int32)
    ops_l, prm_l = [], []
    for i, nd in enumerate(nodes):
        o, q = encode(nd)
        ops_l.append(o); prm_l.append(q)
        op_off[i + 1] = op_off[i] + o.size
        prm_off[i + 1] = prm_off[i] + q.size
    ops = np.concatenate(ops_l) if ops_l else np.zeros(0, dtype=np.uint8)
    prm = np.concatenate(prm_l) if prm_l else np.zeros(0, dtype=np.float64)
    if prm.size == 0:
        prm = np.zeros(1, dtype=np.float64)
    return op_off, np.ascontiguousarray(ops), prm_off, np.ascontiguousarray(prm)


def from_tuple(t) -> Node:
    tag = t[0]
    if tag == "WN": return WhiteNoise(t[1])
    if tag == "C": return Constant(t[1])
    if tag == "LIN": return Linear(*t[1:])
    if tag == "SE": return SquaredExponential(*t[1:])
    if tag == "GE": return GammaExponential(*t[1:])
    if tag == "PER": return Periodic(*t[1:])
    if tag == "+": return Plus(from_tuple(t[1]), from_tuple(t[2]))
    if tag == "*": return Times(from_tuple(t[1]), from_tuple(t[2]))
    if tag == "CP": return ChangePoint(from_tuple(t[1]), from_tuple(t[2]), t[3], t[4])
    raise ValueError(tag)
