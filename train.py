"""Entry point named as the reference's README names it (``python train.py``);
the loop itself is gcn_amd/train.py."""
from gcn_amd.train import main

if __name__ == '__main__':
    main()
