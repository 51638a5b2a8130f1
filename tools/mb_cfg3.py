import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import microbench as mb
mb.run(512, 8192, 32, 17, n_tridiag=16, reps=2)
mb.run(64, 8192, 32, 17, n_tridiag=16, reps=2)
