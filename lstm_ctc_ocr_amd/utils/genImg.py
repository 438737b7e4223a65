"""Offline captcha writer for the test_net directory protocol — /root/reference/lib/utils/genImg.py:20-36:
files named `{idx:08d}_{chars}.png`, read back by test.py:82."""
import os

from .gen import gen_rand, render_captcha


def run(num, path):
    if not os.path.exists(path):
        os.makedirs(path)
    for i in range(num):
        chars = gen_rand()
        render_captcha(chars).save(os.path.join(path, '%08d_%s.png' % (i, chars)))
        if i % 100 == 0:
            print('%d images written' % i)


if __name__ == '__main__':
    import sys
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, sys.argv[2] if len(sys.argv) > 2 else './data/val/')
