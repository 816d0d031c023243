cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ens && cd gpurun_out/ens
timeout -k 10 500 python ../../entrypoints/DDL.py --synthetic --train-size 12000 --set data_train_size=10000 --set random_model_size=4 --set predict_model_size=2 --set epochs=10 --set batch_size=32 --set conv2d_size=1,4 --set dense_size=0,3 --no-plot 2>&1 | tail -24
