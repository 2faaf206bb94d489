"""Weight initialisers with the reference's names (reference gcn/inits.py)."""
import numpy as np

from . import tfcompat as tf


def uniform(shape, scale=0.05, name=None):
    """U(-scale, scale), fp32 (reference gcn/inits.py:5-8)."""
    return tf.Variable(tf.random_uniform(shape, minval=-scale, maxval=scale, dtype=tf.float32), name=name)


def glorot(shape, name=None):
    """Glorot & Bengio uniform, range sqrt(6/(fan_in+fan_out))
    (reference gcn/inits.py:11-15)."""
    r = np.sqrt(6.0 / (shape[0] + shape[1]))
    return tf.Variable(tf.random_uniform(shape, minval=-r, maxval=r, dtype=tf.float32), name=name)


def zeros(shape, name=None):
    return tf.Variable(tf.zeros(shape, dtype=tf.float32), name=name)


def ones(shape, name=None):
    return tf.Variable(tf.ones(shape, dtype=tf.float32), name=name)
