"""Upstream-pygcn training script on the MI355X path (run with cwd = this directory, like the
reference: `python train.py`).

Flag names and defaults are the reference's (pygcn/train.py:36-51; `--hidden` default 16 is the
upstream value kept in the comment at train.py:48).  The epoch body has upstream semantics, which
the fork preserves as comments (train.py:140 `optimizer.zero_grad()`, :150
`output = model(features, adj)`): train on idx_train with NLL loss, Adam(lr, weight_decay),
validate on idx_val, test on idx_test.  The fork's own body (SafeGraph samples, gradient
accumulation, MLP head) is out of scope.
"""
from __future__ import division
from __future__ import print_function

import argparse
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from utils import load_data, accuracy, DEFAULT_CORA
from models import GCN

parser = argparse.ArgumentParser()
parser.add_argument('--no-cuda', action='store_true', default=False,
                    help='Disables CUDA training (unsupported here: the path is HIP-only).')
parser.add_argument('--fastmode', action='store_true', default=False,
                    help='Validate during training pass.')
parser.add_argument('--seed', type=int, default=42, help='Random seed.')
parser.add_argument('--epochs', type=int, default=200, help='Number of epochs to train.')
parser.add_argument('--lr', type=float, default=0.01, help='Initial learning rate.')
parser.add_argument('--weight_decay', type=float, default=5e-4,
                    help='Weight decay (L2 loss on parameters).')
parser.add_argument('--hidden', type=int, default=16, help='Number of hidden units.')
parser.add_argument('--dropout', type=float, default=0.5,
                    help='Dropout rate (1 - keep probability).')
parser.add_argument('--path', default=DEFAULT_CORA,
                    help='Directory with <dataset>.cites[/.content], a .cites file, or the '
                         'committed edge-list fixture.')
parser.add_argument('--dataset', default='cora')


def main():
    args = parser.parse_args()
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("pygcn_amd runs on an MI355X (HIP) device only; there is no CPU path.")
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed)

    adj, features, labels, idx_train, idx_val, idx_test = load_data(args.path, args.dataset)
    model = GCN(nfeat=features.shape[1], nhid=args.hidden,
                nclass=int(labels.max().item()) + 1, dropout=args.dropout)
    optimizer = optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)

    model.cuda()
    features, adj, labels = features.cuda(), adj.cuda(), labels.cuda()
    idx_train, idx_val, idx_test = idx_train.cuda(), idx_val.cuda(), idx_test.cuda()

    def train(epoch):
        t = time.time()
        model.train()
        optimizer.zero_grad()
        output = model(features, adj)
        loss_train = F.nll_loss(output[idx_train], labels[idx_train])
        acc_train = accuracy(output[idx_train], labels[idx_train])
        loss_train.backward()
        optimizer.step()
        if not args.fastmode:
            model.eval()
            output = model(features, adj)
        loss_val = F.nll_loss(output[idx_val], labels[idx_val])
        acc_val = accuracy(output[idx_val], labels[idx_val])
        print('Epoch: {:04d}'.format(epoch + 1),
              'loss_train: {:.4f}'.format(loss_train.item()),
              'acc_train: {:.4f}'.format(acc_train.item()),
              'loss_val: {:.4f}'.format(loss_val.item()),
              'acc_val: {:.4f}'.format(acc_val.item()),
              'time: {:.4f}s'.format(time.time() - t))

    def test():
        model.eval()
        output = model(features, adj)
        loss_test = F.nll_loss(output[idx_test], labels[idx_test])
        acc_test = accuracy(output[idx_test], labels[idx_test])
        print("Test set results:", "loss= {:.4f}".format(loss_test.item()),
              "accuracy= {:.4f}".format(acc_test.item()))

    t_total = time.time()
    for epoch in range(args.epochs):
        train(epoch)
    print("Optimization Finished!")
    print("Total time elapsed: {:.4f}s".format(time.time() - t_total))
    test()


if __name__ == "__main__":
    main()
