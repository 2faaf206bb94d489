"""Trainer: the counterpart of the reference's entry script
(gcn/train_normal.py, the fork's copy of upstream ``train.py``) as a function
over an explicit configuration.

    python train.py --dataset cora --model gcn

The command line keeps the reference's flag names and defaults
(gcn/train_normal.py:18-26), the per-epoch line and the closing lines keep its
print format (:94-96, :103, :106-107), and early stopping follows its rule
(:98-100).  Everything else is organised for this implementation: the inputs
are uploaded once and stay on the GPU, each distinct ``session.run`` is recorded
in a HIP graph and replayed, and a run returns its trace as data.
"""
from __future__ import division, print_function

import collections
import json
import time

import numpy as np

from . import tfcompat as tf
from . import utils

flags = tf.app.flags
FLAGS = flags.FLAGS

# flag name -> (kind, default, help).  The first nine are the reference's.
_FLAG_TABLE = collections.OrderedDict([
    ('dataset', ('string', 'cora', 'Dataset string.')),  # 'cora', 'citeseer', 'pubmed'
    ('model', ('string', 'gcn', 'Model string.')),  # 'gcn', 'gcn_cheby', 'dense'
    ('learning_rate', ('float', 0.01, 'Initial learning rate.')),
    ('epochs', ('integer', 200, 'Number of epochs to train.')),
    ('hidden1', ('integer', 16, 'Number of units in hidden layer 1.')),
    ('dropout', ('float', 0.5, 'Dropout rate (1 - keep probability).')),
    ('weight_decay', ('float', 5e-4, 'Weight for L2 loss on embedding matrix.')),
    ('early_stopping', ('integer', 10, 'Tolerance for early stopping (# of epochs).')),
    ('max_degree', ('integer', 3, 'Maximum Chebyshev polynomial degree.')),
    ('seed', ('integer', 123, 'Random seed (the reference hard-codes 123).')),
    ('report', ('string', '', 'Optional path of a JSON run report.')),
    ('chebyshev_recurrence', ('boolean', False,
                              'gcn_cheby: evaluate sum_k T_k(L) P_k by Clenshaw recurrence over the pattern of L '
                              'instead of the explicit T_k matrices (same polynomial, fp32 rounding differs).')),
    ('device_eigval', ('boolean', False,
                       'gcn_cheby: largest Laplacian eigenvalue by a Lanczos iteration over the HIP SpMM '
                       'instead of ARPACK on the host (about 1e-6 relative difference).')),
    ('hip_graph', ('boolean', True,
                   'Record each distinct session.run in a HIP graph and replay it (this loop feeds the same '
                   'objects every epoch; false = launch every kernel every epoch).')),
])

Config = collections.namedtuple('Config', list(_FLAG_TABLE))


def define_flags():
    """Register the flags once (the library reads ``FLAGS.hidden1`` /
    ``learning_rate`` / ``weight_decay`` as the reference's does)."""
    for name, (kind, default, text) in _FLAG_TABLE.items():
        if name not in FLAGS:
            getattr(flags, 'DEFINE_' + kind)(name, default, text)


def config_from_flags():
    define_flags()
    return Config(**{name: getattr(FLAGS, name) for name in _FLAG_TABLE})


def _publish(cfg):
    """Make ``cfg`` the process-wide FLAGS the model classes read."""
    define_flags()
    for name, value in cfg._asdict().items():
        setattr(FLAGS, name, value)


# --------------------------------------------------------------------- inputs
def supports_for(cfg, adj, largest_eigval=None):
    """``(model class, list of support tuples)`` for ``cfg.model``
    (reference gcn/train_normal.py:33-46)."""
    from .models import GCN, MLP
    if cfg.model == 'gcn_cheby':
        if largest_eigval is None and cfg.device_eigval:
            from .spectral import largest_eigval_of_adjacency
            largest_eigval = largest_eigval_of_adjacency(adj)
        return GCN, utils.chebyshev_polynomials(adj, cfg.max_degree, largest_eigval=largest_eigval)
    if cfg.model == 'gcn':
        return GCN, [utils.preprocess_adj(adj)]
    if cfg.model == 'dense':
        return MLP, [utils.preprocess_adj(adj)]   # fed but unused by the MLP, as in the reference
    raise ValueError('Invalid argument for model: ' + str(cfg.model))


def make_placeholders(n_supports, feature_shape, n_classes):
    """The placeholders dict the layers and models index
    (keys of reference gcn/train_normal.py:49-56)."""
    return {
        'support': [tf.sparse_placeholder(tf.float32) for _ in range(n_supports)],
        'features': tf.sparse_placeholder(tf.float32, shape=tf.constant(feature_shape, dtype=tf.int64)),
        'labels': tf.placeholder(tf.float32, shape=(None, n_classes)),
        'labels_mask': tf.placeholder(tf.int32),
        'dropout': tf.placeholder_with_default(0., shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32),
    }


class Split(object):
    """One labelled subset and the feed dict that selects it; built once so that
    every epoch feeds the same objects (which is what lets a recorded HIP graph
    be replayed)."""

    def __init__(self, features, support, labels, mask, placeholders, dropout=None):
        self.feed = utils.construct_feed_dict(features, support, labels, mask, placeholders)
        if dropout is not None:
            self.feed[placeholders['dropout']] = dropout


def stops_early(val_losses, patience):
    """The reference's rule: past epoch ``patience``, stop when the newest
    validation loss exceeds the mean of the ``patience`` before it."""
    epoch = len(val_losses) - 1
    return epoch > patience and val_losses[-1] > np.mean(val_losses[-(patience + 1):-1])


def _fmt(x):
    return "{:.5f}".format(x)


# ------------------------------------------------------------------------ fit
def fit(cfg, data=None, verbose=True, initial_weights=None, mask_provider=None, largest_eigval=None):
    """Train ``cfg.model`` on ``cfg.dataset`` (or on ``data``, a tuple shaped
    like ``utils.load_data``'s).  ``initial_weights`` ({variable name: array})
    and ``mask_provider`` let a parity run share weights and dropout masks with
    the oracle; ``largest_eigval`` pins the Chebyshev rescaling.  Returns the
    per-epoch trace ``(train loss, train acc, val loss, val acc, seconds)``, the
    test result, the model and the session."""
    _publish(cfg)
    np.random.seed(cfg.seed)
    tf.reset_default_graph()
    tf.set_random_seed(cfg.seed)

    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = \
        data if data is not None else utils.load_data(cfg.dataset)
    features = utils.preprocess_features(features)
    model_class, support = supports_for(cfg, adj, largest_eigval)

    placeholders = make_placeholders(len(support), features[2], y_train.shape[1])
    model = model_class(placeholders, input_dim=features[2][1], logging=True)
    sess = tf.Session(config={"hip_graph": True} if cfg.hip_graph else None)
    sess.mask_provider = mask_provider
    sess.run(tf.global_variables_initializer())
    for name, value in (initial_weights or {}).items():
        model.vars[name].load(value)

    train_split = Split(features, support, y_train, train_mask, placeholders, dropout=cfg.dropout)
    val_split = Split(features, support, y_val, val_mask, placeholders)
    test_split = Split(features, support, y_test, test_mask, placeholders)

    def evaluate(split):
        t0 = time.time()
        loss, acc = sess.run([model.loss, model.accuracy], feed_dict=split.feed)
        return float(loss), float(acc), time.time() - t0

    trace, val_losses = [], []
    for epoch in range(cfg.epochs):
        t0 = time.time()
        _, loss, acc = sess.run([model.opt_op, model.loss, model.accuracy], feed_dict=train_split.feed)
        val_loss, val_acc, _ = evaluate(val_split)
        seconds = time.time() - t0
        val_losses.append(val_loss)
        trace.append((float(loss), float(acc), val_loss, val_acc, seconds))
        if verbose:
            print("Epoch:", '%04d' % (epoch + 1), "train_loss=", _fmt(loss), "train_acc=", _fmt(acc),
                  "val_loss=", _fmt(val_loss), "val_acc=", _fmt(val_acc), "time=", _fmt(seconds))
        if stops_early(val_losses, cfg.early_stopping):
            if verbose:
                print("Early stopping...")
            break
    if verbose:
        print("Optimization Finished!")

    test_cost, test_acc, test_seconds = evaluate(test_split)
    if verbose:
        print("Test set results:", "cost=", _fmt(test_cost), "accuracy=", _fmt(test_acc),
              "time=", _fmt(test_seconds))
    result = {
        "dataset": cfg.dataset, "model": cfg.model, "epochs_run": len(trace),
        "test_cost": test_cost, "test_acc": test_acc,
        "epoch_ms_median": float(np.median([row[4] for row in trace]) * 1e3) if trace else float("nan"),
        "trace": trace, "model_obj": model, "session": sess,
    }
    if cfg.report:
        with open(cfg.report, "w") as fh:
            json.dump({k: v for k, v in result.items() if k not in ("model_obj", "session")}, fh)
    return result


def run(**kwargs):
    """:func:`fit` under the current process-wide FLAGS."""
    return fit(config_from_flags(), **kwargs)


def main():
    define_flags()
    run()


if __name__ == '__main__':
    main()
