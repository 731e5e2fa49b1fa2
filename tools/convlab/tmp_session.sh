cd /root/repo
tools/convlab/mfma_peak
tools/convlab/mfma_peak
