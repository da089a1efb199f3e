﻿!mod$ v1 sum:c23654933239c87c
module poly
contains
subroutine rescal_coeff(lmax,an,bn,cn,rn)
integer(4)::lmax
real(8)::an(0_8:int(lmax-1_4,kind=8))
real(8)::bn(0_8:int(lmax-1_4,kind=8))
real(8)::cn(0_8:int(lmax-1_4,kind=8))
real(8)::rn(0_8:int(lmax,kind=8))
end
subroutine pol2pos(xi,nx,lmax,x,an,bn,cn,cl,p0)
integer(4),intent(in)::nx
real(8),intent(out)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
real(8),intent(in)::an(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cl(0_8:int(lmax,kind=8))
real(8)::p0
end
subroutine pol2pos_omp(xi,nx,lmax,x,an,bn,cn,cl,p0)
integer(4),intent(in)::nx
real(8),intent(out)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
real(8),intent(in)::an(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cl(0_8:int(lmax,kind=8))
real(8)::p0
end
subroutine pol2pos_omp_zsym(xi,nx,lmax,x,an,bn,cn,cl,p0)
integer(4),intent(in)::nx
real(8),intent(out)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
real(8),intent(in)::an(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cl(0_8:int(lmax,kind=8))
real(8)::p0
end
subroutine pol2pos_clshw_omp(xi,nx,lmax,x,an,bn,cn,cl,p0)
integer(4),intent(in)::nx
real(8),intent(out)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
real(8),intent(in)::an(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(lmax-1_4,kind=8))
real(8),intent(in)::cl(0_8:int(lmax,kind=8))
real(8)::p0
end
subroutine pos2pol(xi,nx,lmax,x,an,bn,cn,cl,p0,kmax)
integer(4),intent(in)::nx
real(8),intent(in)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
integer(4),intent(in)::kmax
real(8),intent(in)::an(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(kmax-1_4,kind=8))
real(8),intent(out)::cl(0_8:int(lmax,kind=8))
real(8)::p0(1_8:int(nx,kind=8))
end
subroutine pos2pol_omp(xi,nx,lmax,x,an,bn,cn,cl,p0,kmax)
integer(4),intent(in)::nx
real(8),intent(in)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
integer(4),intent(in)::kmax
real(8),intent(in)::an(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(kmax-1_4,kind=8))
real(8),intent(out)::cl(0_8:int(lmax,kind=8))
real(8)::p0(1_8:int(nx,kind=8))
end
subroutine pos2pol_omp_zsym(xi,nx,lmax,x,an,bn,cn,cl,p0,kmax)
integer(4),intent(in)::nx
real(8),intent(in)::xi(1_8:int(nx,kind=8))
integer(4),intent(in)::lmax
real(8),intent(in)::x(1_8:int(nx,kind=8))
integer(4),intent(in)::kmax
real(8),intent(in)::an(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::bn(0_8:int(kmax-1_4,kind=8))
real(8),intent(in)::cn(0_8:int(kmax-1_4,kind=8))
real(8),intent(out)::cl(0_8:int(lmax,kind=8))
real(8)::p0(1_8:int(nx,kind=8))
end
end
