import json, sys
d = json.load(open(sys.argv[1]))
print('value %.1f img/s  ms/step %.2f' % (d['value'], d['ms_per_step']))
print('roofline', d['roofline'])
print('mfma', d['mfma_kernels'])
names = {1:'CONV',2:'WGRAD',3:'WGRAD_REDUCE',4:'BN_FINALIZE',5:'SUM_TERMS',6:'GRAD_TERM',7:'BN_BWD_REDUCE',8:'BN_BWD_FINALIZE',9:'BILINEAR_CAT',10:'BILINEAR_CAT_BWD',11:'IM2COL',12:'NHWC2NCHW',13:'NCHW2NHWC',15:'BIAS_GRAD',17:'PACK'}
for k, v in d['op_kind_ms'].items(): print('%-16s %5d launches %8.3f ms  %6.2f us avg' % (names.get(int(k), k), v[0], v[1], 1e3*v[1]/v[0]))
for k, v in d['kernel_ms'].items(): print('%5d %8.3f ms %7.2f us  %s' % (v[0], v[1], 1e3*v[1]/v[0], k))
