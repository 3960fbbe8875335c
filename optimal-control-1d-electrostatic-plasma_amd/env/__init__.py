from .pic import PIC
from .batched import BatchedPIC
from .dist import TwoStream, BumpOnTail
from .sharded import ShardedPIC, shard_range
from . import util, interpolate, solve
