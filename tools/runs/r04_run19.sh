cd $GRAFT_REPO_ROOT
tools/ab_multi.sh "p0 pa pb pc pd pe pf" 3 2>&1 | grep variant
