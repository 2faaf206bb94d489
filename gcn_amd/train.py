"""Trainer with the reference's flags, loop, early-stopping rule and print
format (reference gcn/train_normal.py, the fork's copy of upstream train.py).

    python train.py --dataset cora --model gcn
"""
from __future__ import division
from __future__ import print_function

import json
import time

import numpy as np

from . import tfcompat as tf
from .utils import (chebyshev_polynomials, construct_feed_dict, load_data, preprocess_adj,
                    preprocess_features)

flags = tf.app.flags
FLAGS = flags.FLAGS


def define_flags():
    """Names and defaults of reference gcn/train_normal.py:18-26."""
    if 'dataset' in FLAGS:
        return
    flags.DEFINE_string('dataset', 'cora', 'Dataset string.')  # 'cora', 'citeseer', 'pubmed'
    flags.DEFINE_string('model', 'gcn', 'Model string.')  # 'gcn', 'gcn_cheby', 'dense'
    flags.DEFINE_float('learning_rate', 0.01, 'Initial learning rate.')
    flags.DEFINE_integer('epochs', 200, 'Number of epochs to train.')
    flags.DEFINE_integer('hidden1', 16, 'Number of units in hidden layer 1.')
    flags.DEFINE_float('dropout', 0.5, 'Dropout rate (1 - keep probability).')
    flags.DEFINE_float('weight_decay', 5e-4, 'Weight for L2 loss on embedding matrix.')
    flags.DEFINE_integer('early_stopping', 10, 'Tolerance for early stopping (# of epochs).')
    flags.DEFINE_integer('max_degree', 3, 'Maximum Chebyshev polynomial degree.')
    flags.DEFINE_integer('seed', 123, 'Random seed (the reference hard-codes 123).')
    flags.DEFINE_string('report', '', 'Optional path of a JSON run report.')
    flags.DEFINE_boolean('chebyshev_recurrence', False,
                         'gcn_cheby: evaluate sum_k T_k(L) P_k by Clenshaw recurrence over the pattern of L '
                         'instead of the explicit T_k matrices (same polynomial, fp32 rounding differs).')
    flags.DEFINE_boolean('device_eigval', False,
                         'gcn_cheby: largest Laplacian eigenvalue by a Lanczos iteration over the HIP SpMM '
                         'instead of ARPACK on the host (about 1e-6 relative difference).')
    flags.DEFINE_boolean('hip_graph', True, 'Record each distinct session.run in a HIP graph and replay it '
                         '(this loop feeds the same objects every epoch; false = launch every kernel every epoch).')


def run(data=None, verbose=True, initial_weights=None, mask_provider=None):
    """One training run under the current FLAGS.  ``data`` overrides
    ``load_data(FLAGS.dataset)``; ``initial_weights`` ({var name: array}) and
    ``mask_provider`` let a parity run share weights and dropout masks with the
    oracle.  Returns a dict with the per-epoch trace and the test result."""
    from .models import GCN, MLP

    np.random.seed(FLAGS.seed)
    tf.reset_default_graph()
    tf.set_random_seed(FLAGS.seed)

    adj, features, y_train, y_val, y_test, train_mask, val_mask, test_mask = \
        data if data is not None else load_data(FLAGS.dataset)

    features = preprocess_features(features)
    if FLAGS.model == 'gcn':
        support = [preprocess_adj(adj)]
        num_supports = 1
        model_func = GCN
    elif FLAGS.model == 'gcn_cheby':
        lmax = None
        if FLAGS.device_eigval:
            from .spectral import largest_eigval_of_adjacency
            lmax = largest_eigval_of_adjacency(adj)
        support = chebyshev_polynomials(adj, FLAGS.max_degree) if lmax is None else \
            chebyshev_polynomials(adj, FLAGS.max_degree, largest_eigval=lmax)
        num_supports = 1 + FLAGS.max_degree
        model_func = GCN
    elif FLAGS.model == 'dense':
        support = [preprocess_adj(adj)]  # Not used
        num_supports = 1
        model_func = MLP
    else:
        raise ValueError('Invalid argument for model: ' + str(FLAGS.model))

    placeholders = {
        'support': [tf.sparse_placeholder(tf.float32) for _ in range(num_supports)],
        'features': tf.sparse_placeholder(tf.float32, shape=tf.constant(features[2], dtype=tf.int64)),
        'labels': tf.placeholder(tf.float32, shape=(None, y_train.shape[1])),
        'labels_mask': tf.placeholder(tf.int32),
        'dropout': tf.placeholder_with_default(0., shape=()),
        'num_features_nonzero': tf.placeholder(tf.int32)  # helper variable for sparse dropout
    }

    model = model_func(placeholders, input_dim=features[2][1], logging=True)
    sess = tf.Session(config={"hip_graph": True} if FLAGS.hip_graph else None)
    sess.mask_provider = mask_provider

    def evaluate(features, support, labels, mask, placeholders):
        t_test = time.time()
        feed_dict_val = construct_feed_dict(features, support, labels, mask, placeholders)
        outs_val = sess.run([model.loss, model.accuracy], feed_dict=feed_dict_val)
        return outs_val[0], outs_val[1], (time.time() - t_test)

    sess.run(tf.global_variables_initializer())
    if initial_weights:
        for name, value in initial_weights.items():
            model.vars[name].load(value)

    cost_val = []
    trace = []
    for epoch in range(FLAGS.epochs):
        t = time.time()
        feed_dict = construct_feed_dict(features, support, y_train, train_mask, placeholders)
        feed_dict.update({placeholders['dropout']: FLAGS.dropout})

        outs = sess.run([model.opt_op, model.loss, model.accuracy], feed_dict=feed_dict)

        cost, acc, duration = evaluate(features, support, y_val, val_mask, placeholders)
        cost_val.append(cost)
        trace.append((float(outs[1]), float(outs[2]), float(cost), float(acc), time.time() - t))

        if verbose:
            print("Epoch:", '%04d' % (epoch + 1), "train_loss=", "{:.5f}".format(outs[1]),
                  "train_acc=", "{:.5f}".format(outs[2]), "val_loss=", "{:.5f}".format(cost),
                  "val_acc=", "{:.5f}".format(acc), "time=", "{:.5f}".format(time.time() - t))

        if epoch > FLAGS.early_stopping and cost_val[-1] > np.mean(cost_val[-(FLAGS.early_stopping + 1):-1]):
            if verbose:
                print("Early stopping...")
            break

    if verbose:
        print("Optimization Finished!")

    test_cost, test_acc, test_duration = evaluate(features, support, y_test, test_mask, placeholders)
    if verbose:
        print("Test set results:", "cost=", "{:.5f}".format(test_cost),
              "accuracy=", "{:.5f}".format(test_acc), "time=", "{:.5f}".format(test_duration))
    result = {
        "dataset": FLAGS.dataset, "model": FLAGS.model, "epochs_run": len(trace),
        "test_cost": float(test_cost), "test_acc": float(test_acc),
        "epoch_ms_median": float(np.median([r[4] for r in trace]) * 1e3),
        "trace": trace, "model_obj": model, "session": sess,
    }
    if FLAGS.report:
        with open(FLAGS.report, "w") as fh:
            json.dump({k: v for k, v in result.items() if k not in ("model_obj", "session")}, fh)
    return result


def main():
    define_flags()
    run()


if __name__ == '__main__':
    main()
